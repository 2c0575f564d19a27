"""GPU: solver-independent certificates for the results of the HIP path on the variants WITHOUT a recorded
MATLAB/quadprog output -- hard, ondemand, ellip, softall, repair, all3 are **parity-unpinned at the solver boundary**
(SURVEY.md 8c); what these tests add on top of the GPU-vs-oracle comparison is evidence that does not come from an
active-set solver by the same author:

  * every agent the GPU reports SOLVED satisfies the KKT conditions of the reference's literal dense QP
    (stationarity, primal and dual feasibility, complementarity <= 1e-8; multipliers recovered by Lawson-Hanson NNLS);
  * every agent the GPU reports INFEASIBLE -- including the ones rejected by the GPU-only shortcuts (box certificate of
    the scan, dual bound, Farkas test against the acceleration box) -- has an empty constraint set by a phase-1 LP (HiGHS).

Workloads: the bench workload itself (C2, seed SEED0+2, all 512 scenes x 100 agents, solveHardDMPC), C5 repair with
term = -1e6 and -1e7, a C3 sample (1000 agents, soft-all), ondemand / ellip / all3 on the recorded congested scenes, and a
slice of the randomized campaign (tests/dev/gpu_campaign.py) against the oracle.
"""
import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, CAMPAIGN_VARIANTS, load_golden, step14_inputs, init_table
from test_certificates_cpu import check_batch

pytestmark = pytest.mark.gpu


def _report(what, ns, ni, worst):
    print(f"KKT/LP certificate [{what}] (parity unpinned at the solver boundary): {ns} solved agents pass KKT, "
          f"{ni} infeasible agents confirmed by the LP; worst primal {worst['primal']:.1e}, stationarity {worst['stat_rel']:.1e} "
          f"(relative), complementarity {worst['compl_rel']:.1e}, smallest LP infeasibility {worst['t_min']:.2e}")


def _certify_scene(args):
    prm, l, po, z, pf, a, status, tries, what = args
    return check_batch(prm, l, po, z, z, pf, a, status, tries, what)


def test_bench_workload_c2_hard_every_agent_certified():
    """the headline bench workload itself: ALL 512 scenes x 100 agents of bench.py's default run (C2, seed SEED0+2, solveHardDMPC, MPC
    step 2 from the initDMPC table -- every scene aborts there, so this is what bench.py replays): every one of the 51 200 reported
    results carries its certificate -- KKT by NNLS for every solved agent, the phase-1 LP for every infeasible verdict.  (0.6 ms and
    12 ms per agent on the host; the scenes are checked by a pool of forked workers, the GPU is not touched there.)"""
    import multiprocessing as mpx
    import os
    cfg, N, S = wl.CONFIGS["C2"], 100, 512
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
    d = mp.Dmpc("hard", **kw)
    l, _, _ = d.init_batch(po, pf)
    z = np.zeros_like(po)
    out = d.step_batch(l, po, z, z, pf)
    prm = orc.make_params("hard", **kw)
    jobs = [(prm, l[s], po[s], z[s], pf[s], out["a"][s], out["status"][s], out["info"][s][:, 2], f"C2 scene {s}") for s in range(S)]
    try:
        with mpx.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
            res = pool.map(_certify_scene, jobs, chunksize=4)
    except (OSError, ValueError):      # no fork / no pool on this host: the same checks in this process
        res = [_certify_scene(j) for j in jobs]
    tot_s = sum(r[0] for r in res); tot_i = sum(r[1] for r in res)
    worst_all = dict(primal=max(r[2]["primal"] for r in res), stat_rel=max(r[2]["stat_rel"] for r in res),
                     compl_rel=max(r[2]["compl_rel"] for r in res), t_min=min(r[2]["t_min"] for r in res))
    assert tot_s + tot_i == S * N and tot_i > 0.03 * S * N
    _report("C2 hard, the whole bench workload, 51 200 QPs", tot_s, tot_i, worst_all)


@pytest.mark.parametrize("term", [-1e6, -1e7])
def test_c5_repair_certified(term):
    """comp_repair.m:93,194: solveSoftDMPCrepair with term = -1e6 / -1e7 in the dense 200-agent box, three MPC steps"""
    cfg, N = wl.CONFIGS["C5"], 200
    kw = dict(wl.solver_kwargs(cfg, N), term=term)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 5)
    po, pf = po[0], pf[0]
    d = mp.Dmpc("repair", **kw)
    prm = orc.make_params("repair", **kw)
    l = init_table(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(3):
        out = d.step_batch(l, xp, xv, xa, pf)
        viol = np.where(out["info"][:, 0] > 0)[0]
        agents = np.concatenate([viol[:40], np.arange(0, N, 10)])     # the agents with slack rows first
        ns, ni, w = check_batch(prm, l, xp, xv, xa, pf, out["a"], out["status"], out["info"][:, 2], f"C5 term {term:g} step {k + 2}", agents)
        _report(f"C5 repair term {term:g} step {k + 2}", ns, ni, w)
        ok = out["status"] == 1
        l = np.where(ok[:, None], out["p"], l); xp = np.where(ok[:, None], out["p"][:, :3], xp)
        xv = np.where(ok[:, None], out["v"][:, :3], xv); xa = np.where(ok[:, None], out["a"][:, :3], xa)


def test_c3_softall_sample_certified():
    cfg, N = wl.CONFIGS["C3"], 1000
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 3)
    po, pf = po[0], pf[0]
    d = mp.Dmpc("softall", **kw)
    prm = orc.make_params("softall", **kw)
    l = init_table(po, pf)
    z = np.zeros_like(po)
    out = d.step_batch(l, po, z, z, pf)
    viol = np.where(out["info"][:, 0] > 0)[0]
    rng = np.random.default_rng(11)
    agents = np.concatenate([rng.permutation(viol)[:8], rng.integers(0, N, 4)])   # 1044-variable dense QPs: a sample
    ns, ni, w = check_batch(prm, l, po, z, z, pf, out["a"], out["status"], out["info"][:, 2], "C3 softall", agents)
    assert ns >= 10
    _report("C3 softall N=1000 (sample)", ns, ni, w)


@pytest.mark.parametrize("variant", ["ondemand", "ellip", "all3", "softall", "repair", "hard", "cpp1"])
@pytest.mark.parametrize("name", ["failure_rate2_bound", "comp_kctr_3_bound2"])
def test_recorded_congested_scenes_certified(name, variant):
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    out = mp.Dmpc(variant, **kw).step_batch(l, xp, xv, xa, pf)
    prm = orc.make_params(variant, **kw)
    ns, ni, w = check_batch(prm, l, xp, xv, xa, pf, out["a"], out["status"], out["info"][:, 2], f"{name}/{variant}")
    assert ns > 30
    _report(f"{name}/{variant}", ns, ni, w)


def test_campaign_slice():
    """a slice of the randomized parity campaign (tests/dev/gpu_campaign.py): random scenes of 2-90 agents in the C2/C5
    boxes (some shrunk: outbound and infeasible cases), all 10 variants, 2-4 teacher-forced MPC steps each: status, branch
    record and trajectories against the oracle."""
    rng = np.random.default_rng(20180926)
    total = bad = 0
    worst = 0.0
    for it in range(10):
        N = int(rng.integers(2, 90))
        cfgname = "C5" if rng.random() < 0.5 else "C2"
        cfg = wl.CONFIGS[cfgname]
        kw = wl.solver_kwargs(cfg, N)
        if rng.random() < 0.3:
            kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
        po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
        po, pf = po[0], pf[0]
        for variant in ALL_VARIANTS:
            d = mp.Dmpc(variant, **kw); prm = orc.make_params(variant, **kw)
            l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
            tol = 5e-8 if variant in ("softall", "repair", "cpp1", "softall_c") else 1e-9 * max(1.0, abs(kw["term"]) / 5e4)
            for k in range(int(rng.integers(2, 5))):
                out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
                total += N
                ok = (ref["status"] & 1) == 1
                what = f"campaign scene {it} N={N} {cfgname} {variant} step {k + 2}"
                assert np.array_equal(out["status"], ref["status"]), what
                assert np.array_equal(out["info"][:, 0], ref["info"][:, 0]) and np.array_equal(out["info"][:, 1], ref["info"][:, 7]) \
                    and np.array_equal(out["info"][:, 2], ref["info"][:, 2]), what + ": branch record"
                e = max((np.abs(out[key][ok] - ref[key][ok]).max() if ok.any() else 0.0) for key in ("p", "v", "a"))
                worst = max(worst, e)
                assert e <= tol, f"{what}: l_inf {e:.2e}"
                okb = out["status"] & 1 == 1
                l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
                xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
    print(f"campaign slice: {total} agent-steps compared, worst l_inf {worst:.2e}")
    assert total > 5000


def test_campaign_warm_ladder_regressions():
    """the six agent-steps of the full campaign (tests/dev/gpu_campaign.py, seed 1) on which a retry ladder warm-started from the
    factor of an infeasible try went wrong (solveSoftDMPCall: nearly parallel rows leave a nearly degenerate factor): GPU against
    the oracle on exactly those scenes, all MPC steps of their variant."""
    want = {68, 120, 124, 150, 153, 159}
    rng = np.random.default_rng(1)
    seen = 0
    for it in range(max(want) + 1):
        N = int(rng.integers(2, 90))
        cfgname = "C5" if rng.random() < 0.5 else "C2"
        cfg = wl.CONFIGS[cfgname]
        kw = wl.solver_kwargs(cfg, N)
        if rng.random() < 0.3:
            kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
        try:
            po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
        except Exception:
            continue
        po, pf = po[0], pf[0]
        for variant in CAMPAIGN_VARIANTS:
            nst = int(rng.integers(2, 7))            # (the campaign's random stream)
            if it not in want or variant != "all3":
                continue
            d = mp.Dmpc(variant, **kw); prm = orc.make_params(variant, **kw)
            l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
            for k in range(nst):
                out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
                what = f"campaign scene {it} N={N} {cfgname} all3 step {k + 2}"
                assert np.array_equal(out["status"], ref["status"]), what
                assert np.array_equal(out["info"][:, 2], ref["info"][:, 2]), what + ": retry-ladder count"
                ok = (ref["status"] & 1) == 1
                e = max((np.abs(out[key][ok] - ref[key][ok]).max() if ok.any() else 0.0) for key in ("p", "v", "a"))
                assert e <= 1e-9 * max(1.0, abs(kw["term"]) / 5e4), f"{what}: l_inf {e:.2e}"
                okb = out["status"] & 1 == 1
                l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
                xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
            seen += 1
    assert seen == len(want)


def test_campaign_acceptance_floor_regression():
    """campaign seed 73, scene 756 (round 5; found by a 3.9 M agent-step soak, present in the round-4 library too): solveSoftDMPCall in the C5
    box, 62 agents, MPC step 4 -- one agent on retry-ladder level 5 (penalties of 1.6e7) ended 4.8e-4 m off the minimiser with identical status
    and ladder count: a constraint with a refined multiplier of about -5e-3 stayed in the working set because the noise floor of the
    acceptance test was 1e-9 x (1 + largest multiplier) = 0.016.  The floor is 1e-9 + 1e-11 x largest now."""
    rng = np.random.default_rng(73)
    for it in range(757):
        N = int(rng.integers(2, 90))
        cfgname = "C5" if rng.random() < 0.5 else "C2"
        cfg = wl.CONFIGS[cfgname]
        kw = wl.solver_kwargs(cfg, N)
        if rng.random() < 0.3:
            kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
        sc_seed = int(rng.integers(1 << 30))
        nst = {v: int(rng.integers(2, 7)) for v in CAMPAIGN_VARIANTS}     # (the campaign's random stream)
    assert (N, cfgname, nst["all3"]) == (62, "C5", 3)
    po, pf = wl.make_scenes(dict(cfg), 1, N, sc_seed); po, pf = po[0], pf[0]
    d = mp.Dmpc("all3", **kw); prm = orc.make_params("all3", **kw)
    l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(nst["all3"]):
        out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
        what = f"campaign seed 73 scene 756 all3 step {k + 2}"
        assert np.array_equal(out["status"], ref["status"]) and np.array_equal(out["info"][:, 2], ref["info"][:, 2]), what
        ok = (ref["status"] & 1) == 1
        e = max((np.abs(out[key][ok] - ref[key][ok]).max() if ok.any() else 0.0) for key in ("p", "v", "a"))
        assert e <= 1e-9 * max(1.0, abs(kw["term"]) / 5e4), f"{what}: l_inf {e:.2e}"
        okb = out["status"] & 1 == 1
        l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
        xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)


def test_randomized_campaign_time_boxed():
    """One fresh seed of the randomized campaign (tests/dev/gpu_campaign.py: random scenes of 2-90 agents in the C2/C5 boxes, all 10
    variants, 2-6 teacher-forced MPC steps) for 25 s of wall time: every status, branch record and retry count identical to the oracle,
    trajectories to the stated tolerance.  The full campaigns (four seeds, 1.9 M agent-steps) found two real solver defects in round 2;
    this slice keeps a moving sample of them in the collected suite -- the seed changes with the calendar week, and is printed."""
    import time
    seed = int(time.time() // (7 * 86400))
    rng = np.random.default_rng(seed)
    t0 = time.time()
    total = 0
    worst = 0.0
    scenes = 0
    while time.time() - t0 < 25.0:
        N = int(rng.integers(2, 90))
        cfgname = "C5" if rng.random() < 0.5 else "C2"
        cfg = wl.CONFIGS[cfgname]
        kw = wl.solver_kwargs(cfg, N)
        if rng.random() < 0.3:
            kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
        try:
            po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
        except Exception:
            continue
        po, pf = po[0], pf[0]
        scenes += 1
        for variant in ALL_VARIANTS:
            d = mp.Dmpc(variant, **kw); prm = orc.make_params(variant, **kw)
            l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
            tol = 5e-8 if variant in ("softall", "repair", "cpp1", "softall_c") else 1e-9 * max(1.0, abs(kw["term"]) / 5e4)
            for k in range(int(rng.integers(2, 7))):
                out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
                total += N
                what = f"campaign seed {seed} scene {scenes} N={N} {cfgname} {variant} step {k + 2}"
                assert np.array_equal(out["status"], ref["status"]), what
                assert np.array_equal(out["info"][:, 0], ref["info"][:, 0]) and np.array_equal(out["info"][:, 1], ref["info"][:, 7]) \
                    and np.array_equal(out["info"][:, 2], ref["info"][:, 2]), what + ": branch record"
                ok = (ref["status"] & 1) == 1
                ea = np.zeros(N)
                for key in ("p", "v", "a"):
                    ea = np.maximum(ea, np.abs(out[key] - ref[key]).max(axis=1) * ok)
                e = float(ea.max())
                worst = max(worst, e)
                if variant in ("bound", "bound2", "cpp", "cpp2"):
                    # the reduced solver (round 6) forms its small system explicitly: its stated tolerance where the multipliers are huge
                    ladder = ref["info"][:, 2] > 1
                    assert (ea[~ladder] <= max(tol, 5e-8)).all() and (ea[ladder] <= 5e-7).all(), f"{what}: l_inf {e:.2e}"
                else:
                    assert e <= tol, f"{what}: l_inf {e:.2e}"
                okb = out["status"] & 1 == 1
                l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
                xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
    print(f"time-boxed campaign, seed {seed}: {scenes} scenes, {total} agent-steps compared, worst l_inf {worst:.2e}")
    assert total > 1000
