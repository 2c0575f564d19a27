"""GPU: edge cases of the hot path against the oracle -- degenerate scene sizes, agents at the goal, the
`coll` / `outbound` / infeasible branches, random synthetic scenes of the BASELINE configs."""
import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, compare_to_oracle, init_table

pytestmark = pytest.mark.gpu

KW = dict(h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4, pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2))


def _both(variant, l, xp, xv, xa, pf, kw=KW, tol=1e-9):
    out = mp.Dmpc(variant, **kw).step_batch(l, xp, xv, xa, pf)
    ref = orc.step(orc.make_params(variant, **kw), l, xp, xv, xa, pf)
    return out, ref, compare_to_oracle(out, ref, tol, variant)


@pytest.mark.parametrize("variant", ALL_VARIANTS)
def test_single_agent_scene(variant):
    """N = 1: no neighbours at all (empty scan, empty row set, Ain_coll = [])."""
    po, pf = np.array([[0.0, 0.0, 1.0]]), np.array([[2.0, -1.0, 1.5]])
    z = np.zeros((1, 3))
    out, ref, _ = _both(variant, init_table(po, pf), po, z, z, pf)
    assert out["status"][0] == 1 and out["info"][0, 1] == 0 and out["info"][0, 3] == 0   # far case


@pytest.mark.parametrize("variant", ["bound", "bound2", "hard", "ondemand"])
def test_two_agents_head_on_and_near_goal(variant):
    po = np.array([[-1.0, 0.0, 1.0], [1.0, 0.02, 1.0]])
    pf = po[::-1].copy()
    z = np.zeros((2, 3))
    out, ref, _ = _both(variant, init_table(po, pf), po, z, z, pf)
    # agent closer than 1 m to its goal: "near" cost case (solveSoftDMPCbound.m:48-52)
    po2, pf2 = np.array([[0.0, 0.0, 1.0], [2.0, 2.0, 1.0]]), np.array([[0.3, 0.1, 1.0], [1.5, 2.0, 1.2]])
    out2, ref2, _ = _both(variant, init_table(po2, pf2), po2, z, z, pf2)
    assert list(out2["info"][:, 3]) == [1, 1] or variant == "hard"


def test_coll_outbound_and_infeasible_branches():
    # coll: two agents already inside rmin - 0.05 at horizon step 1 (solveSoftDMPCbound.m:25-31)
    po = np.array([[0.0, 0.0, 1.0], [0.2, 0.0, 1.0], [2.0, 2.0, 1.0]])
    pf = np.array([[1.0, 0.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 1.0]])
    z = np.zeros((3, 3))
    out, ref, _ = _both("bound", init_table(po, pf), po, z, z, pf)
    assert out["status"][0] == mp.ST_COLL and out["status"][1] == mp.ST_COLL and out["status"][2] == 1
    # outbound: state outside the workspace moving outwards -> position-bound rows infeasible or first
    # predicted position out of the box (is_inbounds.m)
    po = np.array([[2.49, 0.0, 1.0], [-2.0, 0.0, 1.0]])
    pf = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 1.0]])
    xv = np.array([[1.5, 0.0, 0.0], [0.0, 0.0, 0.0]])
    out, ref, _ = _both("bound", init_table(po, pf), po, xv, z[:2], pf)
    assert out["status"][0] in (mp.ST_INFEAS, mp.ST_SOLVED | mp.ST_OUTBOUND)
    # hard constraints that cannot be met within |a| <= alim: infeasible, single attempt
    po = np.array([[0.0, 0.0, 1.0], [0.36, 0.0, 1.0]])
    pf = np.array([[1.0, 0.0, 1.0], [-1.0, 0.0, 1.0]])
    xv = np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]])
    l = init_table(po, pf)
    out, ref, _ = _both("hard", l, po, xv, z[:2], pf)
    assert np.all(out["status"] == mp.ST_INFEAS) and np.all(out["info"][:, 2] == 1)


@pytest.mark.parametrize("cfg_name,N,steps", [("C2", 100, 3), ("C5", 60, 4), ("C3", 40, 3)])
def test_random_config_scenes_teacher_forced(cfg_name, N, steps):
    """Synthetic scenes of the BASELINE configs (reduced N so the dense oracle stays fast): several
    consecutive MPC steps, each fed identically to GPU and oracle (teacher forcing on the GPU's states)."""
    cfg = wl.CONFIGS[cfg_name]
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 11)
    po, pf = po[0], pf[0]
    d = mp.Dmpc(cfg["variant"], **kw)
    prm = orc.make_params(cfg["variant"], **kw)
    l = init_table(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    tol = 2e-8 if cfg["variant"] in ("softall", "repair", "cpp1") else 1e-9
    for k in range(steps):
        out = d.step_batch(l, xp, xv, xa, pf)
        ref = orc.step(prm, l, xp, xv, xa, pf)
        compare_to_oracle(out, ref, tol, f"{cfg_name} step {k + 2}")
        ok = out["status"] == 1
        l = np.where(ok[:, None], out["p"], l); xp = np.where(ok[:, None], out["p"][:, :3], xp)
        xv = np.where(ok[:, None], out["v"][:, :3], xv); xa = np.where(ok[:, None], out["a"][:, :3], xa)


def test_many_scenes_and_odd_sizes():
    """Ragged sizes: N not a multiple of the wave size, S scenes of different content."""
    cfg = wl.CONFIGS["C4"]
    for N, S in ((7, 5), (65, 3), (129, 2)):
        kw = wl.solver_kwargs(cfg, N)
        po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + N)
        d = mp.Dmpc("bound", **kw)
        l, _, _ = d.init_batch(po, pf)
        z = np.zeros_like(po)
        out = d.step_batch(l, po, z, z, pf)
        prm = orc.make_params("bound", **kw)
        for s in range(S):
            ref = orc.step(prm, l[s], po[s], z[s], z[s], pf[s])
            compare_to_oracle({k: v[s] for k, v in out.items()}, ref, 1e-9, f"N={N} scene {s}")


def test_cpp_flavour_semantics():
    """DMPC::solveQPv2 (dmpc/cpp/dmpc.cpp:803-1287) next to its MATLAB siblings: growing near-neighbour radius
    rmin*(1+(float)k/k_hor), a first-step collision is flagged next to a solution, no in-bounds flag, 21 solves at most.
    No output of that code path is recorded in the reference (parity unpinned): the checker is the oracle's literal
    restatement of the C++ text."""
    f32 = lambda x: float(np.float32(x))
    kwc = dict(KW, h=f32(0.2), rmin=f32(0.35), c=f32(2.0), alim=f32(1.0), term=-1e6)
    # first-step collision: bound returns COLL without outputs, cpp returns a solution AND the flag (dmpc.cpp:419-424)
    po = np.array([[0.0, 0.0, 1.0], [0.2, 0.0, 1.0], [2.0, 2.0, 1.0]])
    pf = np.array([[1.0, 0.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 1.0]])
    z = np.zeros((3, 3))
    l = init_table(po, pf)
    for variant in ("cpp", "cpp2"):
        d = mp.Dmpc(variant, **kwc)
        out = d.step_batch(l, po, z, z, pf)
        ref = orc.step(orc.make_params(variant, **kwc), l, po, z, z, pf)
        compare_to_oracle(out, ref, 1e-9, variant)
        assert out["status"][0] & mp.ST_COLL and out["status"][1] & mp.ST_COLL and out["status"][2] == 1
        assert (out["status"][:2] & (mp.ST_SOLVED | mp.ST_INFEAS)).all()          # solved or proven infeasible, never silently dropped
    # the neighbour radius: a neighbour at 1.5 rmin is a row for `bound` (3 rmin) at every step, for cpp only once
    # rmin (1 + k/15) >= 1.5 rmin, i.e. from k = 8 (0-based) on; violation first seen at horizon step 3 -> no row there
    rm = kwc["rmin"]
    po = np.array([[0.0, 0.0, 1.0], [0.9 * rm, 0.0, 1.0], [0.0, 1.5 * rm, 1.0]])
    pf = po.copy()
    l = np.tile(po[:, None, :], (1, 15, 1)).reshape(3, 45)                  # everybody predicted to stay put
    l3 = l.reshape(3, 15, 3).copy(); l3[1, :2, 0] = 1.2 * rm; l = l3.reshape(3, 45)   # agent 1 only comes close from step 3 on
    rows_b = mp.Dmpc("bound", **kwc).rows_one(l, 0, po[0], z[0])
    rows_c = mp.Dmpc("cpp", **kwc).rows_one(l, 0, po[0], z[0])
    assert rows_b["viol_k"] == rows_c["viol_k"] == 3 and len(rows_b["kc"]) == 2 and len(rows_c["kc"]) == 1
    ref_c = orc.rows_one(orc.make_params("cpp", **kwc), l, 0, po[0], z[0])
    assert ref_c["nrows"] == 1 and np.abs(ref_c["b"][0] - rows_c["rhs"][0]) < 1e-12
    # outbound is never reported by the C++ flavour
    po = np.array([[2.49, 0.0, 1.0], [-2.0, 0.0, 1.0]])
    pf = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 1.0]])
    xv = np.array([[0.3, 0.0, 0.0], [0.0, 0.0, 0.0]])
    out = mp.Dmpc("cpp", **kwc).step_batch(init_table(po, pf), po, xv, z[:2], pf)
    assert not (out["status"] & mp.ST_OUTBOUND).any()


def test_cpp_first_version_semantics():
    """DMPC::solveQP, the FIRST C++ version (dmpc/cpp/dmpc.cpp:554-801; callers solveDMPC :1367, cluster_solve :1776): rows for ALL N-1
    neighbours at the first violating step k, on step k-1 (:480-485), every row with its own unbounded slack (:629-633), one solve,
    no in-bounds flag.  A violation at k = 0 makes the reference index row -3 (undefined there): `coll`, no solve.  No output of this
    code path is recorded in the reference (parity unpinned at the solver boundary): the checkers are the oracle's literal restatement
    and, in tests/test_gpu_certificates.py, the KKT certificate."""
    f32 = lambda x: float(np.float32(x))
    kwc = dict(KW, h=f32(0.2), rmin=f32(0.35), c=f32(2.0), alim=f32(1.0), term=-1e6)
    rm = kwc["rmin"]
    z = np.zeros((3, 3))
    # everybody predicted to stay put; agent 1 comes inside rmin of agent 0 from horizon step 3 (1-based) on, agent 2 stays 1.5 rmin away
    po = np.array([[0.0, 0.0, 1.0], [0.9 * rm, 0.0, 1.0], [0.0, 1.5 * rm, 1.0]])
    pf = po.copy()
    l3 = np.tile(po[:, None, :], (1, 15, 1)); l3[1, :2, 0] = 1.2 * rm
    l = l3.reshape(3, 45)
    rows = mp.Dmpc("cpp1", **kwc).rows_one(l, 0, po[0], z[0])
    ref = orc.rows_one(orc.make_params("cpp1", **kwc), l, 0, po[0], z[0])
    assert rows["viol_k"] == 3 and len(rows["kc"]) == 2 == ref["nrows"]          # BOTH neighbours (solveQPv2 would take one)
    assert (np.asarray(rows["kc"]) == 2).all()                                    # on horizon step k - 1 = 2 (1-based, as viol_k)
    assert np.abs(np.asarray(ref["b"]) - np.asarray(rows["rhs"])).max() < 1e-12
    out = mp.Dmpc("cpp1", **kwc).step_batch(l, po, z, z, pf)
    refs = orc.step(orc.make_params("cpp1", **kwc), l, po, z, z, pf)
    compare_to_oracle(out, refs, 2e-8, "cpp1")
    assert (out["status"] == 1).all() and (out["info"][:, 2] == 1).all()           # one solve each
    # violation at the very first horizon step: undefined in the reference (row -3 of A0) -> coll, no outputs
    po = np.array([[0.0, 0.0, 1.0], [0.2, 0.0, 1.0], [2.0, 2.0, 1.0]])
    pf = np.array([[1.0, 0.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 1.0]])
    l = init_table(po, pf)
    out = mp.Dmpc("cpp1", **kwc).step_batch(l, po, z, z, pf)
    refs = orc.step(orc.make_params("cpp1", **kwc), l, po, z, z, pf)
    compare_to_oracle(out, refs, 2e-8, "cpp1 first-step")
    assert out["status"][0] == mp.ST_COLL and out["status"][1] == mp.ST_COLL and out["status"][2] == 1
    # never outbound
    po = np.array([[2.49, 0.0, 1.0], [-2.0, 0.0, 1.0]])
    pf = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 1.0]])
    xv = np.array([[0.3, 0.0, 0.0], [0.0, 0.0, 0.0]])
    out = mp.Dmpc("cpp1", **kwc).step_batch(init_table(po, pf), po, xv, z[:2], pf)
    assert not (out["status"] & mp.ST_OUTBOUND).any()


@pytest.mark.parametrize("variant", ["softall", "ellip", "repair", "cpp1"])
def test_super_ellipsoid_of_order_4(variant):
    """order = 4 (test/comp_test_ellipconstr.m:158-187 runs solveSoftDMPC with E1 = E^-1, E2 = E^-4): dist = |E1 d|_4, xi = E2 d.^3,
    prev_dist = dist^3 in scan and rows of the all-neighbour variants.  GPU against the oracle (whose order-4 rows are checked against a
    literal numpy restatement of CollConstrEllipDMPC.m in tests/test_oracle_golden.py) on a congested recorded scene and on random C3 / C5
    scenes, several teacher-forced steps; the other variants refuse order 4."""
    from helpers import load_golden, step14_inputs
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g)
    tol = {"ellip": 1e-9, "cpp1": 5e-7}.get(variant, 2e-8)   # (slack penalties of 1e5 / 1e6: the allowance scales with them)
    out = mp.Dmpc(variant, order=4, **kw).step_batch(l, xp, xv, xa, pf)
    ref = orc.step(orc.make_params(variant, order=4, **kw), l, xp, xv, xa, pf, nthreads=8)
    compare_to_oracle(out, ref, tol, f"{variant} order 4 (recorded scene)")
    o2 = mp.Dmpc(variant, **kw).step_batch(l, xp, xv, xa, pf)
    assert (out["info"][:, 1] > 0).any() and not np.array_equal(out["p"], o2["p"])          # rows were built, and they are not the order-2 rows
    cfg = wl.CONFIGS["C5"]; N = 60
    kw5 = wl.solver_kwargs(cfg, N)
    po, pf5 = wl.make_scenes(cfg, 1, N, wl.SEED0 + 77); po, pf5 = po[0], pf5[0]
    d, prm = mp.Dmpc(variant, order=4, **kw5), orc.make_params(variant, order=4, **kw5)
    l5 = init_table(po, pf5); xp5, xv5, xa5 = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(4):
        o = d.step_batch(l5, xp5, xv5, xa5, pf5); r = orc.step(prm, l5, xp5, xv5, xa5, pf5)
        compare_to_oracle(o, r, tol, f"{variant} order 4 C5 step {k + 2}")
        ok = o["status"] == 1
        l5 = np.where(ok[:, None], o["p"], l5); xp5 = np.where(ok[:, None], o["p"][:, :3], xp5)
        xv5 = np.where(ok[:, None], o["v"][:, :3], xv5); xa5 = np.where(ok[:, None], o["a"][:, :3], xa5)
    with pytest.raises(Exception, match="order"):
        mp.Dmpc("bound", order=4, **kw)


@pytest.mark.parametrize("variant", ["softall", "ellip", "repair", "cpp1"])
def test_rows_one_of_order_4_matches_the_oracle(variant):
    """dmpc_rows_one (the standalone a5/a6 entry behind the CollConstr* / CheckColl* shims) on an order-4 context: it launched the order-2
    scan kernel until round 5 (rows with dist = 2-norm and z scaled by c^-4: neither order).  GPU rows against the oracle's, whose order-4
    branches are checked against literal restatements in tests/test_oracle_golden.py."""
    from helpers import load_golden
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv = g["l"], g["pk"][:, 12], g["vk"][:, 12]
    d, prm = mp.Dmpc(variant, order=4, **kw), orc.make_params(variant, order=4, **kw)
    Lam = orc.model_matrices(kw["h"], 15)[0]
    checked = 0
    for n in range(0, l.shape[0], 5):
        r, o = d.rows_one(l, n, xp[n], xv[n]), orc.rows_one(prm, l, n, xp[n], xv[n])
        assert r["nrows"] == o["nrows"] and r["viol_k"] == o["viol_k"], n
        if r["nrows"] == 0:
            continue
        G = np.asarray(o["G"]).reshape(-1, 45)
        for i in range(r["nrows"]):
            kc = r["kc"][i]
            dense = -(r["xi"][i] @ Lam[3 * (kc - 1):3 * kc])
            assert np.abs(dense - G[i]).max() < 1e-12 * max(1.0, np.abs(G[i]).max()), (n, i)
        assert np.abs(r["rhs"] - np.asarray(o["b"])).max() < 1e-11, n
        checked += 1
    assert checked > 3
