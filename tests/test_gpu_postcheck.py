"""GPU: whole-transition post-checks (dmpc_postcheck, failure_rate.m:136-195) against the MATLAB record and the
numpy/scipy oracle (oracle/postcheck.py)."""
import os

import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import driver, workload as wl
from oracle import postcheck as PC
from helpers import GOLD, unrescale

pytestmark = pytest.mark.gpu

KW = dict(h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4, pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2))


def _check(out, s, ref, tol=1e-10):
    assert abs(out["r_factor"][s] - ref["r_factor"]) <= 1e-13 * ref["r_factor"]
    assert abs(out["h_scaled"][s] - ref["h_scaled"]) <= 1e-13
    assert out["n_samples"][s] == ref["n_samples"]
    assert out["min_dist"][s] == ref["min_dist"] or abs(out["min_dist"][s] - ref["min_dist"]) <= tol
    assert out["violation"][s] == ref["violation"]
    assert abs(out["totdist"][s] - ref["totdist"]) <= tol * max(1.0, ref["totdist"])
    assert out["traj_time"][s] == pytest.approx(ref["traj_time"], abs=1e-12)
    if "p" in out:
        n = ref["n_samples"]
        assert np.abs(out["p"][s][:, :n] - ref["p"]).max() <= tol
        assert not out["p"][s][:, n:].any()


def test_postcheck_matches_matlab_record():
    g = np.load(os.path.join(GOLD, "postcheck_comp_kctr_2.npz"))
    p, v, a = unrescale(g)
    d = mp.Dmpc("bound2", **dict(KW, h=float(g["h"]), rmin=float(g["rmin"]), c=float(g["c"])))
    out = d.postcheck([p.shape[1]], g["pf"], p, v, a, vmax=float(g["vmax"]), amax=float(g["amax"]), Ts=float(g["Ts"]), interp=True)
    assert abs(out["r_factor"][0] - float(g["r_factor"])) < 1e-14 and abs(out["h_scaled"][0] - float(g["h_scaled"])) < 1e-14
    assert out["n_samples"][0] == int(g["n_samples"])
    assert np.abs(out["p"][0][:, g["p_idx"]] - g["p"]).max() < 1e-11          # MATLAB's own spline(tk, pk, t)
    assert abs(out["totdist"][0] - float(g["totdist"])) < 1e-9
    assert out["traj_time"][0] == pytest.approx(float(g["traj_time"]), abs=1e-12)
    assert out["violation"][0] == int(g["violation"]) == 0


def _random_hist(rng, N, KT, KTa):
    a = np.zeros((N, KTa, 3)); v = np.zeros_like(a); p = np.zeros_like(a)
    a[:, :KT] = rng.uniform(-1, 1, (N, KT, 3)) * rng.uniform(0.2, 1.0)
    a[:, 0] = 0
    p[:, 0] = rng.uniform(-2, 2, (N, 3))
    for k in range(1, KT):
        v[:, k] = v[:, k - 1] + 0.2 * a[:, k]
        p[:, k] = p[:, k - 1] + 0.2 * v[:, k - 1] + 0.02 * a[:, k]
    return p, v, a


def test_postcheck_ragged_scenes_vs_oracle():
    rng = np.random.default_rng(5)
    S, N, KTa = 5, 7, 40
    used = np.array([40, 23, 4, 31, 12], dtype=np.int32)
    P, V, A = (np.zeros((S, N, KTa, 3)) for _ in range(3))
    for s in range(S):
        P[s], V[s], A[s] = _random_hist(rng, N, used[s], KTa)
    pf = P[np.arange(S), :, used - 1] + rng.normal(0, 0.02, (S, N, 3))
    d = mp.Dmpc("bound", **KW)
    out = d.postcheck(used, pf, P, V, A, interp=True)
    for s in range(S):
        ref = PC.postcheck(P[s][:, :used[s]], V[s][:, :used[s]], A[s][:, :used[s]], pf[s], KW["h"], KW["rmin"], KW["c"])
        _check(out, s, ref)
    # inputs are not modified and the call is repeatable bit for bit
    out2 = d.postcheck(used, pf, P, V, A, interp=True)
    for k in out:
        assert np.array_equal(out[k], out2[k]), k


def test_postcheck_collision_detected():
    """two agents swapping places on a line through each other -> violation; parallel lanes -> none."""
    KT = 30
    d = mp.Dmpc("bound", **KW)
    for lane, expect in ((0.0, 1), (1.0, 0)):
        a = np.zeros((2, KT, 3)); v = np.zeros_like(a); p = np.zeros_like(a)
        p[0, 0] = (-1, 0, 1); p[1, 0] = (1, lane, 1)
        a[0, 1:6, 0], a[1, 1:6, 0] = 0.5, -0.5
        a[0, 20:25, 0], a[1, 20:25, 0] = -0.5, 0.5
        for k in range(1, KT):
            v[:, k] = v[:, k - 1] + 0.2 * a[:, k]
            p[:, k] = p[:, k - 1] + 0.2 * v[:, k - 1] + 0.02 * a[:, k]
        out = d.postcheck([KT], p[:, -1], p, v, a)
        ref = PC.postcheck(p, v, a, p[:, -1], 0.2, 0.35, 2.0)
        assert out["violation"][0] == ref["violation"] == expect
        assert abs(out["min_dist"][0] - ref["min_dist"]) < 1e-10


def test_postcheck_after_transition_resident_equals_host():
    cfg, N, KTm = wl.CONFIGS["C4"], 12, 151
    po, pf = wl.make_scenes(cfg, 3, N=N)
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc(cfg["variant"], **kw)
    tr = d.transition(po, pf, KTm)
    assert (tr["scene_status"] == (mp.ST_SOLVED | mp.ST_REACHED)).all() and (tr["K_T_used"] < KTm).all()
    res = d.postcheck(tr["K_T_used"], pf, KT_alloc=KTm)                         # histories still on the device
    host = d.postcheck(tr["K_T_used"], pf, tr["pk"], tr["vk"], tr["ak"])
    for k in res:
        assert np.array_equal(res[k], host[k]), k
    for s in range(po.shape[0]):
        n = int(tr["K_T_used"][s])
        ref = PC.postcheck(tr["pk"][s][:, :n], tr["vk"][s][:, :n], tr["ak"][s][:, :n], pf[s], kw["h"], kw["rmin"], kw["c"])
        _check(res, s, ref)
        assert ref["violation"] == 0 and ref["r_factor"] > 0


def test_postcheck_edge_cases():
    d = mp.Dmpc("bound", **KW)
    rng = np.random.default_rng(11)
    for N, KT in ((1, 10), (3, 4), (3, 3), (2, 2), (40, 6)):
        p, v, a = _random_hist(rng, N, KT, KT)
        if KT == 2:
            a[:, 0] = 0.3   # failure_rate.m rescales a_1 only; keep r_factor finite
        pf = p[:, -1].copy()
        out = d.postcheck([KT], pf, p, v, a, interp=True)
        ref = PC.postcheck(p, v, a, pf, 0.2, 0.35, 2.0)
        if N == 1:
            assert out["violation"][0] == 0 and np.isinf(out["min_dist"][0])
            ref["min_dist"], ref["violation"] = out["min_dist"][0], 0
        _check(out, 0, ref)
    with pytest.raises(RuntimeError):
        d.postcheck([1], np.zeros((2, 3)), np.zeros((2, 5, 3)), np.zeros((2, 5, 3)), np.zeros((2, 5, 3)))     # K_T_used < 2
    with pytest.raises(RuntimeError):
        d.postcheck([5], np.zeros((2, 3)), np.zeros((2, 5, 3)), np.zeros((2, 5, 3)), np.zeros((2, 5, 3)))     # all-zero histories
    with pytest.raises(RuntimeError):
        mp.Dmpc("bound", **KW).postcheck([5], np.zeros((2, 3)), KT_alloc=5)                                   # nothing resident


def test_postcheck_mask_and_run_trial():
    """failed scenes are skipped (failure_rate.m:136), even when their histories are degenerate; run_trial strings the
    transition and the post-checks together like one trial of the reference's scripts."""
    rng = np.random.default_rng(3)
    S, N, KT = 3, 4, 20
    P, V, A = (np.zeros((S, N, KT, 3)) for _ in range(3))
    for s in (0, 2):
        P[s], V[s], A[s] = _random_hist(rng, N, KT, KT)
    d = mp.Dmpc("bound", **KW)
    pf = P[:, :, -1].copy()
    with pytest.raises(RuntimeError):
        d.postcheck([KT] * S, pf, P, V, A)                       # scene 1 is all zero
    out = d.postcheck([KT, 1, KT], pf, P, V, A, mask=[1, 0, 1])  # ...and may even carry an invalid column count
    assert np.isnan(out["r_factor"][1]) and np.isnan(out["totdist"][1]) and out["n_samples"][1] == 0 and out["violation"][1] == 0
    for s in (0, 2):
        _check(out, s, PC.postcheck(P[s], V[s], A[s], pf[s], 0.2, 0.35, 2.0))
    none = d.postcheck([KT] * S, pf, P, V, A, mask=[0, 0, 0])
    assert np.isnan(none["totdist"]).all()

    cfg, N = wl.CONFIGS["C2"], 10
    po, pf = wl.make_scenes(cfg, 4, N=N)
    po[3, 1] = po[3, 0] + (0.05, 0, 0)                           # scene 3 starts in collision: aborted trial
    kw = wl.solver_kwargs(cfg, N)
    tr = driver.run_trial(mp.Dmpc(cfg["variant"], **kw), po, pf, 151)
    assert not tr["success"][3] and not tr["feasible"][3] and np.isnan(tr["totdist"][3])
    assert tr["success"].any()
    for s in range(4):
        checked = tr["feasible"][s] and not tr["failed_goal"][s]
        assert tr["success"][s] == (checked and tr["violation"][s] == 0)
        if not checked:
            assert np.isnan(tr["totdist"][s]) and np.isnan(tr["traj_time"][s])
            continue
        n = int(tr["K_T_used"][s])
        ref = PC.postcheck(tr["pk"][s][:, :n], tr["vk"][s][:, :n], tr["ak"][s][:, :n], pf[s], kw["h"], kw["rmin"], kw["c"])
        assert abs(tr["totdist"][s] - ref["totdist"]) < 1e-9 and tr["violation"][s] == ref["violation"]
        assert tr["traj_time"][s] == pytest.approx(ref["traj_time"], abs=1e-12) and tr["traj_time"][s] > 0


def test_failure_rate_experiment_statistics():
    """test/failure_rate.m end to end at two swarm sizes: N = 20 always succeeds, N = 100 succeeds in 94 % of the
    reference's 50 recorded trials (failure_rate2.mat) -- 128 random trials here must land in the same region."""
    cfg = wl.CONFIGS["C4"]
    for N, lo, hi in ((20, 1.0, 1.0), (100, 0.80, 0.995)):
        kw = wl.solver_kwargs(cfg, N)
        po, pf = wl.make_scenes(cfg, 128, N, wl.SEED0 + 7 * N)
        res = driver.run_trial(mp.Dmpc("bound", **kw), po, pf, 151, cfg["error_tol"])
        p = res["success"].mean()
        assert lo <= p <= hi, (N, p)
        assert not res["failed_goal"].any()                      # the reference's failures are infeasibility / collisions, not timeouts
        assert (res["success"] == (res["feasible"] & ~res["failed_goal"] & (res["violation"] == 0))).all()


def test_failure_rate_curve_against_the_reference_record(capsys):
    """The loop level pinned to the reference's own outcome record: test/failure_rate.m:61-203 at ALL ten swarm sizes N = 20 .. 200, 200 random
    trials each here (randomTest scenes from the device generator, density-scaled box, failure_rate.m constants), against the 50 trials per
    size MATLAB recorded in data/failure_rate/failure_rate2.mat (tests/golden/failure_rate2_outcomes.npz: success_dmpc, feasible, violation,
    coll per trial; 49 completed trials at N = 200).  What is compared is `success` = feasible && ~failed_goal && ~violation (:196): the
    failure KINDS are not comparable one to one -- after a `coll` return the reference's loop carries on with a zero-filled history
    (failure_rate.m:112-126 breaks only the agent loop) and such a trial ends as infeasible or as a violation of the
    post-check, here it ends at the collision.  Two-proportion z-scores under the pooled binomial; the bar is |z| <= 3.3 per size (seeds are
    fixed, so this is a regression bar, not a coin) and the table with the 95 % Wilson intervals is printed (pytest -s / the captured output).
    The dense end (N >= 120) is where an exact QP solver and quadprog at ConstraintTolerance = 1e-3 (solveSoftDMPCbound.m:10-13) could part
    ways: they do not -- which is the evidence for NOT building an acceptance-slack "reference loop" mode."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "failure_rate2_outcomes.npz"))
    done = g["completed"].astype(bool)
    cfg = wl.CONFIGS["C4"]
    S = 200

    def wilson(k, n, z=1.96):
        p = k / n
        den = 1 + z * z / n
        c, hw = (p + z * z / (2 * n)) / den, z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / den
        return c - hw, c + hw
    rows, worst = [], 0.0
    for qi, N in enumerate(int(x) for x in g["N_vector"]):
        kw = wl.solver_kwargs(cfg, N)
        d = mp.Dmpc("bound", **kw)
        po, pf = wl.make_scenes_device(d, cfg, S, N, wl.SEED0 + 7 * N)
        res = driver.run_trial(d, po, pf, cfg["K_T"], cfg["error_tol"], histories=False)
        k1, n1 = int(res["success"].sum()), S
        n0 = int(done[qi].sum()); k0 = int((g["success_dmpc"][qi].astype(bool) & done[qi]).sum())
        pp = (k0 + k1) / (n0 + n1)
        z = 0.0 if pp in (0.0, 1.0) else (k1 / n1 - k0 / n0) / np.sqrt(pp * (1 - pp) * (1 / n0 + 1 / n1))
        worst = max(worst, abs(z))
        st = res["scene_status"]
        kinds = dict(infeasible=int(((st & mp.ST_INFEAS) != 0).sum()), coll=int(((st & mp.ST_COLL) != 0).sum()), outbound=int(((st & mp.ST_OUTBOUND) != 0).sum()),
                     violation=int(res["violation"].sum()), failed_goal=int(res["failed_goal"].sum()))
        rows.append((N, k1 / n1, wilson(k1, n1), k0 / n0, wilson(k0, n0), z, kinds,
                     dict(feasible=int((g["feasible"][qi].astype(bool) & done[qi]).sum()), violation=int((g["violation"][qi].astype(bool) & done[qi]).sum()),
                          coll=int((g["coll"][qi].astype(bool) & done[qi]).sum()), n=n0)))
        assert not res["failed_goal"].any()        # (the reference records none either: trials fail by infeasibility / collision, not by running out of steps)
    with capsys.disabled():
        print("\n   N   success here (200 trials, 95 % Wilson)   recorded (MATLAB, 95 % Wilson)        z     failures here / recorded flags")
        for N, p1, w1, p0, w0, z, kinds, rec in rows:
            print(f" {N:3d}   {p1:.3f} [{w1[0]:.3f}, {w1[1]:.3f}]                  {p0:.3f} [{w0[0]:.3f}, {w0[1]:.3f}] (n = {rec['n']})    {z:+5.2f}   {kinds} / {rec}")
    assert worst <= 3.3, rows


def test_run_trial_without_history_download_gives_the_same_outcomes():
    cfg = wl.CONFIGS["C4"]
    N = 16
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 5, N, wl.SEED0 + 21)
    a = driver.run_trial(mp.Dmpc("bound", **kw), po, pf, 151, cfg["error_tol"])
    b = driver.run_trial(mp.Dmpc("bound", **kw), po, pf, 151, cfg["error_tol"], histories=False)
    assert b["pk"] is None and a["pk"] is not None
    for k in ("K_T_used", "scene_status", "success", "feasible", "violation", "totdist", "traj_time", "r_factor"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


# ---- large scenes: the cell-grid search (N > 256; failure_rate.m:170-181 is O(N^2) per sample) -------------------------------

def _swarm_hist(rng, N, KT, side, speed=1.0):
    """N agents spread over a cube of the given side (C4 density at side = N^(1/3)), smooth random motion for KT knots"""
    a = rng.uniform(-1, 1, (N, KT, 3)) * speed
    a[:, 0] = 0
    v = np.zeros_like(a); p = np.zeros_like(a)
    m = int(np.ceil(N ** (1.0 / 3.0)))     # jittered lattice: no two starts closer than 0.6 lattice steps
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[rng.permutation(m ** 3)[:N]]
    p[:, 0] = (g + 0.5 + rng.uniform(-0.2, 0.2, (N, 3))) * (side / m) + (-side / 2, -side / 2, 0.2)
    for k in range(1, KT):
        v[:, k] = v[:, k - 1] + 0.2 * a[:, k]
        p[:, k] = p[:, k - 1] + 0.2 * v[:, k - 1] + 0.02 * a[:, k]
    return p, v, a


def _kw_for(N):
    s = N ** (1.0 / 3.0)
    return dict(KW, pmin=(-s / 2, -s / 2, 0.2), pmax=(s / 2, s / 2, s + 0.2))


@pytest.mark.parametrize("N,KT", [(300, 9), (1000, 7), (10000, 5)])
def test_postcheck_large_scene_grid_vs_oracle(N, KT):
    """N = 10^3 and 10^4 (BASELINE configs C3/C4 sizes): min_dist / violation of the device's cell-grid search against the
    oracle's k-d-tree search (oracle/postcheck.py: min_dist_tree, itself checked against the literal pair loop), everything else
    against the literal restatement; then a planted near-miss and a planted collision between two far-apart indices."""
    rng = np.random.default_rng(100 + N)
    kw = _kw_for(N)
    p, v, a = _swarm_hist(rng, N, KT, N ** (1.0 / 3.0))
    pf = p[:, -1] + rng.normal(0, 0.02, (N, 3))
    d = mp.Dmpc("bound", **kw)
    for plant in (None, 0.31, 0.2):
        if plant is not None:   # agent N-1 flies next to agent 3: same history, offset `plant` along x (ellipsoidal distance = plant)
            p[N - 1] = p[3] + (plant, 0, 0); v[N - 1] = v[3]; a[N - 1] = a[3]
        out = d.postcheck([KT], pf, p, v, a)
        ref = PC.postcheck(p, v, a, pf, kw["h"], kw["rmin"], kw["c"], pairs="tree")
        assert abs(out["min_dist"][0] - ref["min_dist"]) <= 1e-10, (N, plant, out["min_dist"][0], ref["min_dist"])
        assert out["violation"][0] == ref["violation"]
        assert out["n_samples"][0] == ref["n_samples"] and abs(out["r_factor"][0] - ref["r_factor"]) <= 1e-13 * ref["r_factor"]
        assert abs(out["totdist"][0] - ref["totdist"]) <= 1e-10 * ref["totdist"]
        assert out["traj_time"][0] == pytest.approx(ref["traj_time"], abs=1e-12)
        if plant is not None:   # (the planted pair keeps its offset at every sample; some other pair may come closer still)
            assert out["min_dist"][0] <= plant + 1e-9
            if plant < kw["rmin"] - 0.05:
                assert out["violation"][0] == 1


def test_postcheck_grid_equals_brute_force_bit_for_bit():
    """the same scene through the brute-force search (N <= 256: two half scenes padded apart) and through the grid (N = 400):
    the pair that realises the minimum lies in one half, so both searches must return the SAME double"""
    rng = np.random.default_rng(77)
    KT = 8
    kw = dict(KW, pmin=(-4, -4, 0.2), pmax=(12, 4, 4.2))
    pA, vA, aA = _swarm_hist(rng, 200, KT, 2.5)
    pB, vB, aB = _swarm_hist(rng, 200, KT, 2.5)
    pB = pB + (8.0, 0, 0)                      # the halves never come near each other
    d = mp.Dmpc("bound", **kw)
    p, v, a = (np.concatenate(x) for x in ((pA, pB), (vA, vB), (aA, aB)))
    whole = d.postcheck([KT], p[:, -1], p, v, a)
    # the halves alone: r_factor differs per scene, so give each half the whole scene's limits by rescaling against the same extreme agent
    halves = []
    for q, w, b in ((pA, vA, aA), (pB, vB, aB)):
        # append the globally extreme agent far away so that r_factor (a min over the scene) is the whole scene's
        rf_agent = int(np.argmin(np.minimum(1.0 / np.sqrt((a ** 2).sum(-1) + 1e-300), 2.0 / np.sqrt((v ** 2).sum(-1) + 1e-300)).min(1)))
        qq = np.concatenate((q, p[rf_agent:rf_agent + 1] + (0, 30.0, 0))); ww = np.concatenate((w, v[rf_agent:rf_agent + 1])); bb = np.concatenate((b, a[rf_agent:rf_agent + 1]))
        halves.append(d.postcheck([KT], qq[:, -1], qq, ww, bb))
    assert halves[0]["r_factor"][0] == whole["r_factor"][0] == halves[1]["r_factor"][0]
    assert min(halves[0]["min_dist"][0], halves[1]["min_dist"][0]) == whole["min_dist"][0]


def test_postcheck_sparse_large_scene_falls_back_to_brute_force():
    """no pair within the grid's cell edge (2 rmin): 'no violation' is proven by the grid, the exact min_dist comes from the
    brute-force pass over the same sample batches"""
    N, KT = 343, 6
    g = np.stack(np.meshgrid(*[np.arange(7) * 3.0] * 3, indexing="ij"), -1).reshape(-1, 3) + (-9.0, -9.0, 0.5)
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, (N, KT, 3)) * 0.5; a[:, 0] = 0
    v = np.zeros_like(a); p = np.zeros_like(a); p[:, 0] = g
    for k in range(1, KT):
        v[:, k] = v[:, k - 1] + 0.2 * a[:, k]
        p[:, k] = p[:, k - 1] + 0.2 * v[:, k - 1] + 0.02 * a[:, k]
    kw = dict(KW, pmin=(-10, -10, 0.2), pmax=(10, 10, 20.0))
    out = mp.Dmpc("bound", **kw).postcheck([KT], p[:, -1], p, v, a)
    ref = PC.postcheck(p, v, a, p[:, -1], kw["h"], kw["rmin"], kw["c"])
    assert ref["min_dist"] > 2 * kw["rmin"] and out["violation"][0] == 0
    assert abs(out["min_dist"][0] - ref["min_dist"]) <= 1e-10
