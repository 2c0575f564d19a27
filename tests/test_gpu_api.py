"""GPU: the MATLAB-signature mirror (multiagent_planning_amd.api), the device transition loop and the
device-pointer entry points."""
import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import api, driver, workload as wl
from oracle import oracle as orc
from helpers import load_golden, oracle_params, step14_inputs, init_table

pytestmark = pytest.mark.gpu


def _matlab_args(g, kw, n):
    l = g["l"].reshape(-1, 15, 3).transpose(2, 1, 0)   # 3 x K x N
    A, Av, A0, Dl = mp.model_matrices(kw["h"])
    E1 = np.diag([1, 1, 1 / kw["c"]]); E2 = np.diag([1, 1, 1 / kw["c"] ** 2])
    return (g["pk"][n, 12], g["pf"][n], g["vk"][n, 12], g["ak"][n, 12], n + 1, kw["h"], l, 15, kw["rmin"], kw["pmin"], kw["pmax"],
            kw["alim"], A, A0, A, Av, Dl, kw["Q1"], kw["S1"], E1, E2, 2)


def test_solveSoftDMPCbound_signature_and_golden():
    g, kw = load_golden("failure_rate2_bound")
    for n in (0, 1, 6, 50):
        p, v, a, feasible, outbound, coll = api.solveSoftDMPCbound(*_matlab_args(g, kw, n), kw["term"])
        assert p.shape == (3, 15) and (feasible, outbound, coll) == (1, 0, 0)
        assert np.abs(p.T.reshape(-1) - g["new_l"][n]).max() < 2e-6          # MATLAB/quadprog record
        assert np.abs(a[:, 0] - g["ak"][n, 13]).max() < 1e-4
    # the agent at which the recorded trial stopped: coll = 1, feasible = 1, empty outputs (:25-31)
    p, v, a, feasible, outbound, coll = api.solveSoftDMPCbound(*_matlab_args(g, kw, int(g["n_done"])), kw["term"])
    assert p.size == 0 and v.size == 0 and a.size == 0 and (feasible, coll) == (1, 1)


def test_other_signatures():
    g, kw = load_golden("comp_kctr_3_bound2")
    args = _matlab_args(g, kw, 3)
    r = api.solveSoftDMPCbound2(*args, kw["term"])
    assert len(r) == 6 and r[0].shape == (3, 15) and r[3] == 1
    assert np.abs(r[0].T.reshape(-1) - g["new_l"][3]).max() < 2e-6
    assert len(api.solveHardDMPC(*args)) == 6 and len(api.solveHardDMPCOnDemand(*args)) == 6
    assert len(api.solveSoftDMPCall(*args, kw["term"])) == 6
    short = args[:14] + args[16:]   # legacy 20-arg form without A_p, A_v
    assert len(api.solveSoftDMPC(*short)) == 5 and len(api.solveEllipDMPC(*short)) == 5
    assert len(api.solveSoftDMPCrepair(*short, kw["term"])) == 6


def test_transition_c1_reaches_goal_and_matches_python_loop():
    """C1: the 4-agent diagonal swap of dmpc_soft_bound.m; device loop == host loop over step_batch."""
    cfg = wl.CONFIGS["C1"]
    kw = wl.solver_kwargs(cfg)
    po, pf = wl.make_scenes(cfg, 1)
    d = mp.Dmpc(cfg["variant"], **kw)
    res = driver.run_transition(d, po, pf, cfg["K_T"], cfg["error_tol"])
    KT = int(res["K_T_used"][0])
    assert res["scene_status"][0] == (mp.ST_SOLVED | mp.ST_REACHED) and 20 < KT <= cfg["K_T"]
    pk = res["pk"][0]
    assert np.linalg.norm(pk[:, KT - 1] - pf[0], axis=1).max() < cfg["error_tol"]   # ReachedGoal.m
    # no pair ever closer than rmin - 0.05 in ellipsoidal norm at the MPC knots
    e1 = np.array([1, 1, 1 / cfg["c"]])
    for k in range(KT):
        dd = np.sqrt((((pk[:, None, k] - pk[None, :, k]) * e1) ** 2).sum(-1)) + 10 * np.eye(4)
        assert dd.min() > cfg["rmin"] - 0.05
    # host loop replay (teacher-free, deterministic kernel -> bitwise identical)
    l, _, _ = d.init_batch(po[0], pf[0])
    xp, xv, xa = po[0].copy(), np.zeros((4, 3)), np.zeros((4, 3))
    for k in range(1, KT):
        out = d.step_batch(l, xp, xv, xa, pf[0])
        assert np.all(out["status"] == 1)
        l, xp, xv, xa = out["p"], out["p"][:, :3], out["v"][:, :3], out["a"][:, :3]
        assert np.array_equal(xp, pk[:, k]) and np.array_equal(xa, res["ak"][0][:, k])
    # and against the oracle driven closed-loop (sparse scene: no solver-noise divergence expected)
    prm = orc.make_params(cfg["variant"], **kw)
    l = init_table(po[0], pf[0]); xp, xv, xa = po[0].copy(), np.zeros((4, 3)), np.zeros((4, 3))
    for k in range(1, KT):
        o = orc.step(prm, l, xp, xv, xa, pf[0])
        l, xp, xv, xa = o["p"], o["p"][:, :3], o["v"][:, :3], o["a"][:, :3]
    assert np.abs(xp - pk[:, KT - 1]).max() < 1e-7


def test_sharded_device_layout_bitwise_identical():
    """G = 1 vs G = 2, 4 chunked table layouts on one GPU give identical bits."""
    import torch
    g, kw = load_golden("failure_rate2_bound")
    l, xp, xv, xa, pf = step14_inputs(g)
    N, S = 200, 2
    dev = torch.device("cuda", 0)
    rows = np.stack([l, l[::-1]])
    XP, XV, XA, PF = (np.stack([a, a[::-1]]) for a in (xp, xv, xa, pf))
    d = mp.Dmpc("bound", **kw)
    ref = None
    for G in (1, 2, 4):
        C = N // G
        lT = torch.from_numpy(driver.rows_to_chunked(rows, G)).to(dev)
        ps, sts = [], []
        for r in range(G):
            sl = slice(r * C, (r + 1) * C)
            loc = driver.GpuLocalStep(d, S, G, C, dev)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, sl])).to(dev)
            out = loc(lT, t(XP), t(XV), t(XA), t(PF), r)
            torch.cuda.synchronize()
            ps.append(out["p"].cpu().numpy()); sts.append(out["status"].cpu().numpy())
        p = np.concatenate(ps, axis=1); st = np.concatenate(sts, axis=1)
        if ref is None:
            ref = (p, st)
            one = d.step_batch(rows, XP, XV, XA, PF)
            assert np.array_equal(one["p"], p) and np.array_equal(one["status"], st)
        else:
            assert np.array_equal(ref[0], p) and np.array_equal(ref[1], st)


def test_transition_batch_matches_host_loop_and_oracle_outcomes():
    """S scenes x N agents of the primary variant through the device loop (dmpc_transition) == host loop over
    step_batch, bit for bit, including the per-scene stop rule (goal reached / abort)."""
    cfg = dict(wl.CONFIGS["C4"])
    N, S, KT = 24, 3, 60
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 99)
    d = mp.Dmpc("bound", **kw)
    res = d.transition(po, pf, KT, cfg["error_tol"])
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    used = np.full(S, KT); done = np.zeros(S, bool); sst = np.ones(S, int)
    for k in range(1, KT):
        out = d.step_batch(l, xp, xv, xa, pf)
        ok = (out["status"] & 1) == 1
        l = np.where(ok[..., None], out["p"], l)
        xp = np.where(ok[..., None], out["p"][..., :3], xp); xv = np.where(ok[..., None], out["v"][..., :3], xv)
        xa = np.where(ok[..., None], out["a"][..., :3], xa)
        for s in range(S):
            if done[s]:
                continue
            assert np.array_equal(res["pk"][s][:, k], xp[s]), (s, k)
            bits = int(np.bitwise_or.reduce(out["status"][s]))
            if bits & ~1:
                done[s], used[s], sst[s] = True, k + 1, bits
            elif np.linalg.norm(xp[s] - pf[s], axis=1).max() < cfg["error_tol"]:
                done[s], used[s], sst[s] = True, k + 1, mp.ST_SOLVED | mp.ST_REACHED
    assert np.array_equal(res["K_T_used"], used) and np.array_equal(res["scene_status"], sst)
    assert ((res["scene_status"] & ~mp.ST_REACHED) == mp.ST_SOLVED).any()   # at least one scene runs without a failing agent


def test_transition_outcomes_n20():
    """Outcome equivalence (SURVEY.md App. C parity protocol item 4): on sparse 20-agent scenes every transition
    reaches its goals within error_tol, never leaves the workspace, and keeps the ellipsoidal separation
    >= rmin - 0.05 at every MPC knot (the reference's acceptance tests, failure_rate.m:125,170-181)."""
    cfg = dict(wl.CONFIGS["C4"])
    N, S = 20, 6
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 20)
    res = mp.Dmpc("bound", **kw).transition(po, pf, 151, cfg["error_tol"])
    e1 = np.array([1, 1, 1 / cfg["c"]])
    assert np.all(res["scene_status"] == (mp.ST_SOLVED | mp.ST_REACHED))
    for s in range(S):
        KT = int(res["K_T_used"][s])
        pk = res["pk"][s][:, :KT]
        assert 30 < KT < 151
        assert np.linalg.norm(pk[:, -1] - pf[s], axis=1).max() < cfg["error_tol"]
        assert np.all(pk >= np.array(kw["pmin"]) - 0.05) and np.all(pk <= np.array(kw["pmax"]) + 0.05)
        assert np.abs(res["ak"][s][:, :KT]).max() <= cfg["alim"] + 1e-9
        d = np.sqrt((((pk[:, None] - pk[None]) * e1) ** 2).sum(-1)) + 10 * np.eye(N)[:, :, None]
        assert d.min() > cfg["rmin"] - 0.05
        # recorded states obey the reference's dynamics identity p_{k+1} = p_k + h v_k + h^2/2 a_{k+1}
        vk, ak, h = res["vk"][s][:, :KT], res["ak"][s][:, :KT], cfg["h"]
        assert np.abs(pk[:, 1:] - (pk[:, :-1] + h * vk[:, :-1] + h * h / 2 * ak[:, 1:])).max() < 1e-12


def test_small_helpers_match_m_files():
    """initDMPC.m, is_inbounds.m, propStatedmpc.m, dec-iSCP/propState.m, ReachedGoal.m in their standalone (device) form."""
    p, v, a = api.initDMPC([0, 0, 1], [10, 0, 1], 0.2, 15, 101)
    assert p.shape == (3, 15) and np.allclose(p[0], np.arange(15) * 0.2) and (p[2] == 1).all() and not v.any() and not a.any()
    assert api.is_inbounds([2.54, 0, 1], [-2.5, -2.5, 0.2], [2.5, 2.5, 2.2])
    assert not api.is_inbounds([2.56, 0, 1], [-2.5, -2.5, 0.2], [2.5, 2.5, 2.2])
    assert not api.is_inbounds(np.array([[0, 0], [0, 0], [1, 0.14]]), [-2.5, -2.5, 0.2], [2.5, 2.5, 2.2])     # second point below the floor
    A_p, A_v, A0 = api.getModelMats(0.2, 15)
    rng = np.random.default_rng(0)
    acc, po, vo = rng.normal(size=45), rng.normal(size=3), rng.normal(size=3)
    pp, vv = api.propStatedmpc(po, vo, acc, A0, A_p, A_v)
    assert np.abs(pp - (A_p @ acc + A0 @ np.r_[po, vo])).max() < 1e-13          # propStatedmpc.m:3
    assert np.abs(vv - (A_v @ acc + np.tile(vo, 15))).max() < 1e-13             # propStatedmpc.m:4
    K = 16                                                                      # dec-iSCP: A_p is 3(K-1) x 3(K-1)... here 45 x 45
    p2, v2 = api.propState(po, acc, A_p, A_v, K)
    assert p2.shape == (48,) and np.array_equal(p2[:3], po) and not v2[:3].any()
    assert np.abs(p2[3:] - (A_p @ acc + np.tile(po, K - 1))).max() < 1e-13 and np.abs(v2[3:] - A_v @ acc).max() < 1e-13
    pk = np.zeros((3, 5, 2)); pk[:, 4, 0] = [1, 1, 1]; pk[:, 4, 1] = [2, 2, 2]
    pf = np.array([[1, 1, 1.005], [2, 2, 2]]).T.reshape(1, 3, 2)
    assert api.ReachedGoal(pk, pf, 5, 0.01, 2) and not api.ReachedGoal(pk, pf, 5, 0.001, 2)
    assert api.ReachedGoal(pk[:, :, 0], pf[:, :, 0], 5, 0.01, 1)


def test_sharded_transition_loop_on_device_matches_dmpc_transition():
    """driver.run_transition_sharded with the HIP local step (world of one, torch CUDA tensors) walks the same closed loop
    as dmpc_transition: same number of MPC steps and bit-identical histories."""
    import torch
    cfg = wl.CONFIGS["C4"]
    N, S, KT = 12, 3, 120
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 77)
    d = mp.Dmpc("bound", **kw)
    ref = d.transition(po, pf, KT, cfg["error_tol"])
    dev = torch.device("cuda", 0)
    l, _, _ = d.init_batch(po, pf)
    lT = torch.from_numpy(driver.rows_to_chunked(l, 1)).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    stepper = driver.ShardedStepper(driver.GpuLocalStep(d, S, 1, N, dev), 0, 1)
    res = driver.run_transition_sharded(stepper, lT, t(po), t(np.zeros_like(po)), t(np.zeros_like(po)), t(pf), KT, cfg["error_tol"])
    assert np.array_equal(res["K_T_used"], ref["K_T_used"])
    assert np.array_equal(res["reached"], (ref["scene_status"] & mp.ST_REACHED) != 0)
    pk = res["pk"].cpu().numpy()
    for s in range(S):
        n = int(ref["K_T_used"][s])
        assert np.array_equal(pk[s][:, :n], ref["pk"][s][:, :n])


@pytest.mark.parametrize("S", [40, 288])
def test_transition_batches_of_every_depth_equal_single_scene_runs(S):
    """dmpc_transition picks its launch form by the depth of the batch (one-agent workgroups with one working-set tier for
    shallow batches, persistent waves with a ticket queue and a second tier for deep ones, split halves from 32 scenes on);
    whatever it picks, a scene must come out exactly as when it runs alone.  288 x 100 agents sits between the two regimes."""
    cfg = wl.CONFIGS["C4"]
    N = 100
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 55)
    d = mp.Dmpc("bound", **kw)
    big = d.transition(po, pf, 151, cfg["error_tol"])
    assert ((big["scene_status"] & mp.ST_REACHED) != 0).mean() > 0.8
    for s in (0, S // 2, S - 1):
        one = mp.Dmpc("bound", **kw).transition(po[s:s + 1], pf[s:s + 1], 151, cfg["error_tol"])
        assert int(one["K_T_used"][0]) == int(big["K_T_used"][s]) and int(one["scene_status"][0]) == int(big["scene_status"][s]), s
        u = int(one["K_T_used"][0])      # (columns past the stop hold the frozen state in a batch, zeros in a single run)
        assert np.array_equal(one["pk"][0][:, :u], big["pk"][s][:, :u]) and np.array_equal(one["ak"][0][:, :u], big["ak"][s][:, :u]), s


def test_reached_goal_is_checked_on_the_first_column_too():
    """agents that start at their goals: ReachedGoal holds on the initDMPC column (failure_rate.m:125 tests it after k = 1),
    the transition ends with one column and no solve"""
    kw = wl.solver_kwargs(wl.CONFIGS["C4"], 20)
    po, _ = wl.make_scenes(wl.CONFIGS["C4"], 2, 20, wl.SEED0 + 91)
    pf = po.copy()
    pf[1, 3] += 0.5        # scene 1 has one agent with somewhere to go
    for fn in ("transition", "transition_sharded"):
        r = getattr(mp.Dmpc("bound", **kw), fn)(po, pf, 40, 0.01)
        assert int(r["K_T_used"][0]) == 1 and int(r["scene_status"][0]) == (mp.ST_SOLVED | mp.ST_REACHED), fn
        assert int(r["K_T_used"][1]) > 1, fn
