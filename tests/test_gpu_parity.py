"""GPU: parity of the HIP path (through the C ABI) against the CPU oracle and the MATLAB goldens.

Protocol (SURVEY.md Appendix C "Parity protocol"): teacher-forced single MPC steps -- identical
(state, table) inputs on both sides; bar: identical branch records (status, first violating step,
row count, cost case, retry count) and trajectory l_inf(p,v,a) <= 1e-9 (fp64) against the oracle;
two-tier tolerance against the quadprog records.
"""
import numpy as np
import pytest

import multiagent_planning_amd as mp
from oracle import oracle as orc
from helpers import ALL_VARIANTS, load_golden, oracle_params, init_table, step14_inputs, compare_to_oracle

pytestmark = pytest.mark.gpu

TOL = {"softall": 2e-8, "repair": 2e-8, "cpp1": 2e-8, "softall_c": 2e-8}   # |term| = 1e5-scale multipliers: see DESIGN.md section 6


@pytest.mark.parametrize("name,variant", [("failure_rate2_bound", "bound"), ("comp_kctr_3_bound2", "bound2")])
def test_golden_step14(name, variant):
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    nd = int(g["n_done"])
    with_gpu = mp.Dmpc(variant, **kw)
    out = with_gpu.step_batch(l, xp, xv, xa, pf)
    ref = orc.step(oracle_params(variant, kw), l, xp, xv, xa, pf)
    compare_to_oracle(out, ref, 1e-9, name)
    err = np.abs(out["p"][:nd] - g["new_l"][:nd]).max(axis=1)
    assert (err <= 2e-6).sum() >= int(0.75 * nd)
    assert np.median(err) < 1e-8
    assert out["status"][nd] == mp.ST_COLL


@pytest.mark.parametrize("variant", ALL_VARIANTS)
@pytest.mark.parametrize("name", ["failure_rate2_bound", "comp_kctr_3_bound2"])
def test_all_variants_vs_oracle_on_recorded_scenes(name, variant):
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    out = mp.Dmpc(variant, **kw).step_batch(l, xp, xv, xa, pf)
    ref = orc.step(oracle_params(variant, kw), l, xp, xv, xa, pf)
    assert not np.any(out["status"] & (mp.ST_CAPACITY | mp.ST_ITERCAP))
    compare_to_oracle(out, ref, TOL.get(variant, 1e-9), f"{name}/{variant}")


@pytest.mark.parametrize("name,variant", [("failure_rate2_bound", "bound"), ("comp_kctr_3_bound2", "bound2")])
def test_step2_from_init(name, variant):
    g, kw = load_golden(name)
    N = int(g["N"])
    d = mp.Dmpc(variant, **kw)
    l0, v0, a0 = d.init_batch(g["po"], g["pf"])
    assert np.array_equal(l0, init_table(g["po"], g["pf"]))
    z = np.zeros((N, 3))
    out = d.step_batch(l0, g["po"], z, z, g["pf"])
    ref = orc.step(oracle_params(variant, kw), l0, g["po"], z, z, g["pf"])
    compare_to_oracle(out, ref, 1e-9, name + "/step2")
    ea = np.abs(out["a"][:, :3] - g["ak"][:, 1]).max(axis=1)
    assert (ea <= 1e-5).sum() >= N - 2


def test_solve_one_matches_batch():
    g, kw = load_golden("failure_rate2_bound")
    l, xp, xv, xa, pf = step14_inputs(g)
    d = mp.Dmpc("bound", **kw)
    out = d.step_batch(l, xp, xv, xa, pf)
    for n in (0, 1, 3, 10, 157, 169):
        r = d.solve_one(l, n, xp[n], xv[n], xa[n], pf[n])
        assert r["status"] == out["status"][n]
        assert np.array_equal(r["p"], out["p"][n]) and np.array_equal(r["a"], out["a"][n])


def test_scene_batching_is_independent():
    """S scenes in one launch == S separate launches, bit for bit."""
    g1, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g1)
    rng = np.random.default_rng(3)
    perm = rng.permutation(l.shape[0])
    L2 = np.stack([l, l[perm]]); XP = np.stack([xp, xp[perm]]); XV = np.stack([xv, xv[perm]])
    XA = np.stack([xa, xa[perm]]); PF = np.stack([pf, pf[perm]])
    d = mp.Dmpc("bound2", **kw)
    both = d.step_batch(L2, XP, XV, XA, PF)
    one = d.step_batch(l, xp, xv, xa, pf)
    assert np.array_equal(both["p"][0], one["p"]) and np.array_equal(both["status"][0], one["status"])
    # permuting the agents of a scene permutes the answers (neighbour order only affects tie breaks)
    assert np.array_equal(both["status"][1], one["status"][perm])
    assert np.abs(both["p"][1] - one["p"][perm]).max() < 1e-9


@pytest.mark.parametrize("variant", ["bound", "hard", "all3"])
def test_replicated_scenes_are_bitwise_identical(variant):
    """The same scene 24x in one launch: every copy must give identical bits (catches any dependence on
    uninitialised LDS / scheduling; LDS contents differ between first and later workgroups of a CU)."""
    g, kw = load_golden("failure_rate2_bound")
    l, xp, xv, xa, pf = step14_inputs(g)
    S = 24
    rep = lambda a: np.ascontiguousarray(np.broadcast_to(a, (S,) + a.shape))
    out = mp.Dmpc(variant, **kw).step_batch(rep(l), rep(xp), rep(xv), rep(xa), rep(pf))
    for k in ("p", "v", "a", "status", "info"):
        assert np.array_equal(out[k], np.broadcast_to(out[k][0], out[k].shape)), k


@pytest.mark.parametrize("variant", ALL_VARIANTS)
@pytest.mark.parametrize("name", ["failure_rate2_bound", "comp_kctr_3_bound2"])
def test_scan_and_rows_match_oracle(name, variant):
    """a5/a6 on their own: first violating step, `coll`, and every collision row (dense Ain, bin, prev_dist)
    of the GPU scan kernel against the oracle's literal CollConstr* restatement, same row order."""
    from multiagent_planning_amd import api
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    prm = oracle_params(variant, kw)
    l3 = l.reshape(-1, 15, 3).transpose(2, 1, 0)
    E1 = np.diag([1, 1, 1 / kw["c"]]) if variant != "scp" else np.eye(3)   # (solveDMPC: plain Euclidean norm)
    N = l.shape[0]
    for n in list(range(0, N, 9)) + [int(g["n_done"])]:
        ref = orc.rows_one(prm, l, n, xp[n], xv[n])
        Ain, bin_, dist, vk, coll = api.collision_rows(variant, xp[n], xv[n], n + 1, kw["h"], l3, 15, kw["rmin"], kw["pmin"], kw["pmax"],
                                                      kw["alim"], kw["Q1"], kw["S1"], E1, 2, kw["term"])
        assert vk == ref["viol_k"] and coll == int(bool(ref["status"] & 4)), (n, vk, ref["viol_k"])
        assert Ain.shape[0] == ref["nrows"], (n, Ain.shape, ref["nrows"])
        if ref["nrows"]:
            assert np.abs(Ain - ref["G"]).max() <= 1e-14 and np.abs(bin_ - ref["b"]).max() <= 1e-13
            assert np.abs(dist - ref["dist"]).max() <= 1e-14


def test_checkcoll_helper_matches_scan():
    from multiagent_planning_amd import api
    g, kw = load_golden("failure_rate2_bound")
    l3 = g["l"].reshape(-1, 15, 3).transpose(2, 1, 0)
    E1 = np.diag([1, 1, 1 / kw["c"]])
    viol, mind, vc = api.CheckCollSoftDMPC(l3[:, 8, 1], l3, 2, 9, E1, kw["rmin"], 2)
    r = orc.rows_one(oracle_params("bound", kw), g["l"], 1, g["pk"][1, 12], g["vk"][1, 12])
    assert r["viol_k"] == 9 and viol.any() and int(vc.sum()) == r["nrows"]


@pytest.mark.parametrize("variant", ["hard", "bound", "softall"])
def test_persistent_and_per_agent_solve_kernels_are_bitwise_identical(variant, monkeypatch):
    """The solve phase has two launch forms (one agent per workgroup / persistent waves claiming agents from a queue
    with the cost tables shared per workgroup); the choice depends on the launch size only and must not change a bit."""
    import multiagent_planning_amd as mp
    from multiagent_planning_amd import workload as wl
    cfg = wl.CONFIGS["C2"]
    N, S = 60, 9
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 17)
    outs = []
    # third form: two capacity tiers for the slack variants (32 slots first, the agents that outgrow them re-solved with 64)
    for opts in ({"no_persist": 1}, {"force_persist": 1}, {"force_persist": 1, "tier1_qcap": 32}):
        d = mp.Dmpc(variant, **kw)
        for k_, v_ in opts.items():
            d.debug_option(k_, v_)            # development options of the context (dmpc_debug_option): launch forms, never arithmetic
        l, _, _ = d.init_batch(po, pf)
        z = np.zeros_like(po)
        outs.append(d.step_batch(l, po, z, z, pf))
    for k in ("p", "v", "a", "status", "info"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    # two tiers: an agent that outgrows the first tier is solved again from the start of its ladder level (no warm start
    # across the hand-off), so its path -- not its minimiser -- differs: same status, trajectories to solver accuracy
    assert np.array_equal(outs[0]["status"], outs[2]["status"])
    for k in ("p", "v", "a"):
        assert np.abs(outs[0][k] - outs[2][k]).max() <= 1e-9, k
    # the tier hand-off re-solves an agent from the start of its ladder level: same branch record, more iterations counted
    assert np.array_equal(outs[0]["info"][..., :4], outs[2]["info"][..., :4])
    assert (outs[0]["status"] & 1).any()


def test_launch_order_is_pure_scheduling():
    """The solve launch order (order_kernel / its scan-time key) must not change a single bit: the same batch is solved in
    the built-in order, in a random order and in reverse, through the development hook dmpc_debug_set_order."""
    import ctypes as C
    from multiagent_planning_amd import workload as wl
    cfg, N, S = wl.CONFIGS["C2"], 100, 8
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 5)
    d = mp.Dmpc("hard", **kw)
    l, _, _ = d.init_batch(po, pf)
    z = np.zeros_like(po)
    ref = d.step_batch(l, po, z, z, pf)
    L = d._L
    L.dmpc_debug_set_order.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.dmpc_debug_set_order.restype = C.c_int
    T = S * N
    for order in (np.random.default_rng(2).permutation(T), np.arange(T)[::-1]):
        o = np.ascontiguousarray(order, dtype=np.int32)
        assert L.dmpc_debug_set_order(d._ctx, o.ctypes.data_as(C.POINTER(C.c_int)), T) == 0
        out = d.step_batch(l, po, z, z, pf)
        for k in ("p", "v", "a", "status", "info"):
            assert np.array_equal(out[k], ref[k]), k
    assert L.dmpc_debug_set_order(d._ctx, None, 0) == 0


def test_cpp_flavour_against_the_recorded_200_agent_dump():
    """f-4: the HIP path of DMPC::solveQPv2 (variant cpp) against the reference's own recorded output,
    dmpc/cpp_results/trajectories (200-agents).txt (first solve of all 200 agents; constants of that revision:
    tests/test_oracle_golden.py), and against the oracle on the same inputs."""
    from test_oracle_golden import _cpp_dump, check_cpp_dump_first_solve
    g, kw = _cpp_dump()
    N = int(g["N"])
    z = np.zeros((N, 3))
    d = mp.Dmpc("cpp", **kw)
    l, _, _ = d.init_batch(g["po"], g["pf"])
    out = d.step_batch(l, g["po"], z, z, g["pf"])
    assert (out["status"] == 1).all()
    check_cpp_dump_first_solve(out["a"][:, :3], out["info"][:, 0], g, "GPU cpp")
    ref = orc.step(orc.make_params("cpp", **kw), l, g["po"], z, z, g["pf"], nthreads=8)
    compare_to_oracle(out, ref, 1e-9, "cpp dump")
