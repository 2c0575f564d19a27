"""The reduced solver of solveSoftDMPCbound (csrc/dmpc_rsolve.hip, round 6) against the general solver (development option reduced_solver = 0)
and the oracle: same branch records, same minimiser; the hand-over of agents it does not take; workspace walls."""
import os

import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import load_golden, step14_inputs

pytestmark = pytest.mark.gpu


def _c4_like(N, seed):
    cfg = dict(wl.CONFIGS["C4"]); cfg["N"] = N
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 1, N, seed)
    return cfg, kw, po[0], pf[0]


def _agree(out, ref, tol_plain=1e-9, tol_ladder=5e-8):
    st_o, st_r = out["status"].ravel(), ref["status"].ravel()
    assert np.array_equal(st_o, st_r)
    io, ir = out["info"].reshape(-1, 8), ref["info"].reshape(-1, 8)
    assert np.array_equal(io[:, 0], ir[:, 0]) and np.array_equal(io[:, 3], ir[:, 3])
    assert np.array_equal(io[:, 2], ir[:, 2]), "retry-ladder counts"
    ok = (st_r & 1) == 1
    worst = 0.0
    for key in ("p", "v", "a"):
        e = np.abs(out[key].reshape(-1, 45)[ok] - ref[key].reshape(-1, 45)[ok]).max(axis=1)
        first = ir[ok, 2] == 1
        assert (e[first] <= tol_plain).all(), (key, float(e[first].max()))
        assert (e <= tol_ladder).all(), (key, float(e.max()))
        worst = max(worst, float(e.max()))
    return worst


def test_reduced_solver_is_what_runs_for_solveSoftDMPCbound():
    g, kw = load_golden("failure_rate2_bound")
    d = mp.Dmpc("bound", **kw)
    d.step_batch(*step14_inputs(g))
    assert d.last_solve_kernel == "dmpc_rsolve_persist_kernel"
    d.debug_option("reduced_solver", 0)
    d.step_batch(*step14_inputs(g))
    assert d.last_solve_kernel.startswith("dmpc_solve_")
    h = mp.Dmpc("hard", **kw)
    h.step_batch(*step14_inputs(g))
    assert h.last_solve_kernel.startswith("dmpc_solve_")


def test_reduced_against_general_and_oracle_closed_loop():
    """a scene at the headline's density, teacher-forced by the oracle over MPC steps 2-6: every agent of every step"""
    cfg, kw, po, pf = _c4_like(1500, wl.SEED0 + 606)
    prm = orc.make_params("bound", **kw)
    red, gen = mp.Dmpc("bound", **kw), mp.Dmpc("bound", **kw)
    gen.debug_option("reduced_solver", 0)
    l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(len(po))])
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(2, 7):
        ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=os.cpu_count())
        o_r, o_g = red.step_batch(l, xp, xv, xa, pf), gen.step_batch(l, xp, xv, xa, pf)
        _agree(o_r, ref)
        assert np.array_equal(o_r["status"], o_g["status"]) and np.array_equal(o_r["info"][..., :4], o_g["info"][..., :4])
        ok = (ref["status"] & 1) == 1
        l = np.where(ok[:, None], ref["p"], l)
        xp = np.where(ok[:, None], ref["p"][:, :3], xp); xv = np.where(ok[:, None], ref["v"][:, :3], xv); xa = np.where(ok[:, None], ref["a"][:, :3], xa)


def test_level_skip_extrapolation_changes_no_retry_count():
    """the retry ladder skips the levels a failed solve's Farkas combination still proves infeasible (margin 1e-7 of its cancelling sums): with the
    extrapolation switched off (development option no_level_skip: every level the certificate does not settle is solved) the retry counts, the
    statuses and the minimisers are the same -- in the reduced and in the general solver, on MPC step 2 of a dense scene (ladder levels up to 6)"""
    cfg, kw, po, pf = _c4_like(3000, wl.SEED0 + 611)
    l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(len(po))])
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for reduced in (1, 0):
        a, b = mp.Dmpc("bound", **kw), mp.Dmpc("bound", **kw)
        a.debug_option("reduced_solver", reduced); b.debug_option("reduced_solver", reduced)
        b.debug_option("no_level_skip", 1)
        oa, ob = a.step_batch(l, xp, xv, xa, pf), b.step_batch(l, xp, xv, xa, pf)
        assert oa["info"][..., 2].max() >= 3, "the scene must climb the ladder"
        worst = _agree(oa, ob)
        assert worst <= 5e-8
        assert ob["info"][..., 4].sum() >= oa["info"][..., 4].sum()   # (equal when the level certificate settles every level the extrapolation would have skipped)


def test_wall_leaving_while_a_row_enters_regression():
    """scene 452 of the randomized campaign with seed 602 (tests/dev/gpu_campaign.py; 64 agents in the small dense box, MPC step 4, agent 24): a workspace
    wall left the working set during the inner iteration of an entering hard row -- the row moved down one lane of the small system and the cached
    gather of the hard rows' data, keyed on the rows alone, served row 0's there: a "solved" agent 0.97 off the minimiser, violating the row"""
    from helpers import ALL_VARIANTS, init_table
    rng = np.random.default_rng(602)
    for it in range(453):
        N = int(rng.integers(2, 90))
        cfg = wl.CONFIGS["C5" if rng.random() < 0.5 else "C2"]
        kw = wl.solver_kwargs(cfg, N)
        if rng.random() < 0.3:
            kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
        sc_seed = int(rng.integers(1 << 30))
        for _ in ALL_VARIANTS: rng.integers(2, 7)
    assert N == 64
    po, pf = wl.make_scenes(dict(cfg), 1, N, sc_seed); po, pf = po[0], pf[0]
    d, prm = mp.Dmpc("bound", **kw), orc.make_params("bound", **kw)
    l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(2, 7):
        out, ref = d.step_batch(l, xp, xv, xa, pf), orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
        _agree(out, ref)
        ok = (ref["status"] & 1) == 1
        l = np.where(ok[:, None], ref["p"], l)
        xp = np.where(ok[:, None], ref["p"][:, :3], xp); xv = np.where(ok[:, None], ref["v"][:, :3], xv); xa = np.where(ok[:, None], ref["a"][:, :3], xa)


def test_tight_workspace_campaign_time_boxed():
    """a time-boxed slice (15 s) of tests/dev/gpu_campaign_walls.py: scenes at the headline's density squeezed into boxes of 0.5-0.8 of their size, the
    four variants of the reduced solver, 4-8 teacher-forced MPC steps: walls enter and leave the working set while collision rows do (the regime of
    the gather defect of round 6).  Statuses, branch records, retry counts identical; 1e-8 on the first ladder level, 1e-7 above"""
    import time
    rng = np.random.default_rng(int(time.time() // (7 * 86400)) + 17)
    t0, total, walls = time.time(), 0, 0
    while time.time() - t0 < 15.0:
        N = int(rng.integers(30, 160))
        cfg = dict(wl.CONFIGS["C4"]); cfg["N"] = N
        kw = dict(wl.solver_kwargs(cfg, N))
        po, pf = wl.make_scenes(cfg, 1, N, int(rng.integers(1 << 30))); po, pf = po[0], pf[0]
        s = 0.5 + 0.3 * rng.random()
        kw["pmin"] = tuple(np.array(kw["pmin"]) * s + np.array([0, 0, 0.2 * (1 - s)])); kw["pmax"] = tuple(np.array(kw["pmax"]) * s)
        lo, hi = np.array(kw["pmin"]) + 0.02, np.array(kw["pmax"]) - 0.02
        po, pf = np.clip(po * s, lo, hi), np.clip(pf * s, lo, hi)
        for variant in ("bound", "bound2", "cpp", "cpp2"):
            d, prm = mp.Dmpc(variant, **kw), orc.make_params(variant, **kw)
            l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(N)])
            xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
            for k in range(int(rng.integers(4, 9))):
                out, ref = d.step_batch(l, xp, xv, xa, pf), orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
                what = f"N={N} scale {s:.3f} {variant} step {k + 2}"
                assert np.array_equal(out["status"], ref["status"]), what
                assert np.array_equal(out["info"][:, 0], ref["info"][:, 0]) and np.array_equal(out["info"][:, 2], ref["info"][:, 2]), what
                ok = (ref["status"] & 1) == 1
                e = np.zeros(N)
                for key in ("p", "v", "a"):
                    e = np.maximum(e, np.abs(out[key] - ref[key]).max(axis=1) * ok)
                first = ref["info"][:, 2] == 1
                assert (e[first] <= 1e-8).all() and (e <= 1e-7).all(), f"{what}: {e.max():.2e}"
                rel = ref["p"].reshape(-1, 15, 3)[ok]
                walls += int(((np.abs(rel - np.array(kw["pmax"])) < 1e-9) | (np.abs(rel - np.array(kw["pmin"])) < 1e-9)).any(axis=(1, 2)).sum())
                total += N
                l = np.where(ok[:, None], ref["p"], l)
                xp = np.where(ok[:, None], ref["p"][:, :3], xp); xv = np.where(ok[:, None], ref["v"][:, :3], xv); xa = np.where(ok[:, None], ref["a"][:, :3], xa)
    print(f"tight-workspace campaign: {total} agent-steps, {walls} of them with a horizon step on a wall")
    assert total > 5000 and walls > 100


def test_hand_over_to_the_general_solver_changes_nothing():
    """agents the reduced solver gives up (here: nearly all, by a cap of three equality solves per ladder level) are solved by the general
    solver in the tier-2 launch: the step's outputs are then the general solver's, bit for bit, for those agents -- and the oracle's minimiser"""
    cfg, kw, po, pf = _c4_like(600, wl.SEED0 + 607)
    prm = orc.make_params("bound", **kw)
    red, gen = mp.Dmpc("bound", **kw), mp.Dmpc("bound", **kw)
    red.debug_option("rsolve_cap", 3)
    gen.debug_option("reduced_solver", 0)
    l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(len(po))])
    z = np.zeros_like(po)
    o_r, o_g = red.step_batch(l, po, z, z, pf), gen.step_batch(l, po, z, z, pf)
    ref = orc.step(prm, l, po, z, z, pf, nthreads=os.cpu_count())
    _agree(o_r, ref)
    handed = o_r["info"][:, 4] == o_g["info"][:, 4]          # (the general solver's iteration count in the record)
    assert handed.sum() > 100
    same = np.all(o_r["a"] == o_g["a"], axis=1)
    assert same[handed & (o_g["info"][:, 4] > 4)].all()


def test_walls_of_a_tight_workspace():
    """a box so small that agents are pressed against its walls (up to three of them in a corner): the walls are extras of the small system"""
    cfg, kw, po, pf = _c4_like(120, wl.SEED0 + 608)
    s = 0.62
    kw = dict(kw); kw["pmin"] = tuple(np.array(kw["pmin"]) * s + np.array([0, 0, 0.2 * (1 - s)])); kw["pmax"] = tuple(np.array(kw["pmax"]) * s)
    lo, hi = np.array(kw["pmin"]) + 0.02, np.array(kw["pmax"]) - 0.02
    po, pf = np.clip(po * s, lo, hi), np.clip(pf * s, lo, hi)
    prm = orc.make_params("bound", **kw)
    red = mp.Dmpc("bound", **kw)
    l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(len(po))])
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    walls = 0
    for k in range(2, 9):
        ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=os.cpu_count())
        out = red.step_batch(l, xp, xv, xa, pf)
        _agree(out, ref)
        ok = (ref["status"] & 1) == 1
        rel = ref["p"].reshape(-1, 15, 3)[ok]
        walls += int(((np.abs(rel - np.array(kw["pmax"])) < 1e-9) | (np.abs(rel - np.array(kw["pmin"])) < 1e-9)).any(axis=(1, 2)).sum())
        l = np.where(ok[:, None], ref["p"], l)
        xp = np.where(ok[:, None], ref["p"][:, :3], xp); xv = np.where(ok[:, None], ref["v"][:, :3], xv); xa = np.where(ok[:, None], ref["a"][:, :3], xa)
    assert walls > 0, "the scene was meant to press agents against the walls"
