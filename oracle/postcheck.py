"""CPU ORACLE for the whole-transition post-checks (TEST INFRASTRUCTURE ONLY).

Literal numpy restatement of test/failure_rate.m:136-195 (same block in test/comp_kctr.m:274-335): rescale to
the velocity / acceleration limits, 100 Hz spline interpolation, pairwise ellipsoidal collision check, path
length, trajectory time.  MATLAB's `spline` (not-a-knot cubic; base MATLAB, not in /root/reference) is restated
with scipy.interpolate.CubicSpline(bc_type="not-a-knot"), the same interpolant; it is pinned against the
reference's recorded `p = spline(tk, pk, t)` arrays in tests/golden/postcheck_comp_kctr_2.npz
(tests/test_oracle_golden.py).
"""
import numpy as np
from scipy.interpolate import CubicSpline


def scale_factor(vk, ak, vmax=2.0, amax=1.0):
    """failure_rate.m:138-145.  vk, ak: [N, KT, 3].  Returns (r_factor, ak_mod[KT,N], vk_mod[KT,N])."""
    with np.errstate(divide="ignore"):
        ak_mod = (amax / np.sqrt((np.asarray(ak) ** 2).sum(-1))).T
        vk_mod = (vmax / np.sqrt((np.asarray(vk) ** 2).sum(-1))).T
    return min(ak_mod.min(), vk_mod.min()), ak_mod, vk_mod


def sample_times(KT, h_scaled, Ts=0.01):
    """failure_rate.m:149-152: T = (k-2) h_scaled with k-1 = KT recorded columns; tk = 0:h_scaled:T; t = 0:Ts:T."""
    T = (KT - 1) * h_scaled
    ns = int(np.floor(T / Ts + 1e-10)) + 1
    return np.arange(KT) * h_scaled, np.arange(ns) * Ts


def rescale(pk, vk, ak, r_factor, h_scaled):
    """failure_rate.m:156-162 (in place on copies)."""
    pk, vk, ak = (np.array(x, dtype=float, copy=True) for x in (pk, vk, ak))
    for k in range(pk.shape[1] - 1):
        ak[:, k] = ak[:, k] * r_factor
        vk[:, k + 1] = vk[:, k] + h_scaled * ak[:, k]
        pk[:, k + 1] = pk[:, k] + h_scaled * vk[:, k] + h_scaled ** 2 / 2 * ak[:, k]
    return pk, vk, ak


def min_dist_tree(p, c):
    """The minimum of failure_rate.m:170-181 over all pairs and samples WITHOUT the O(N^2) loop: per sample a k-d tree on the
    scaled points E1 p (scipy.spatial.cKDTree, nearest other point of every point).  Same quantity by an unrelated method --
    what the large-scene tests compare the device's cell-grid search with (checked against the literal loop on small scenes,
    tests/test_oracle_golden.py)."""
    from scipy.spatial import cKDTree
    e1 = np.array([1.0, 1.0, 1.0 / c])
    best = np.inf
    if p.shape[0] < 2:
        return best
    for s in range(p.shape[1]):
        q = p[:, s, :] * e1
        d, _ = cKDTree(q).query(q, k=2)
        best = min(best, float(d[:, 1].min()))
    return best


def interp_check(pk, h_scaled, pf, rmin, c, Ts=0.01, pairs="literal"):
    """failure_rate.m:165-194 on rescaled knots pk [N,KT,3].  pairs="tree": the pairwise minimum by min_dist_tree (large scenes)."""
    N, KT, _ = pk.shape
    tk, t = sample_times(KT, h_scaled, Ts)
    if KT >= 4:
        p = np.moveaxis(CubicSpline(tk, np.moveaxis(pk, 1, 0), axis=0, bc_type="not-a-knot")(t), 0, 1)   # [N, ns, 3]  (:165)
    else:
        p = np.stack([CubicSpline(tk, pk[i], axis=0, bc_type="not-a-knot")(t) for i in range(N)])
    e1 = np.array([1.0, 1.0, 1.0 / c])
    min_dist = np.inf
    if pairs == "tree":
        min_dist = min_dist_tree(p, c)
    else:
        for i in range(N):                                      # :170-181
            d = np.sqrt((((p[i][None] - p) * e1) ** 2).sum(-1))
            d[i] = np.inf
            min_dist = min(min_dist, d.min())
    totdist = float(np.sqrt((np.diff(p, axis=1) ** 2).sum(-1)).sum())   # :183
    dist_goal = np.sqrt(((p - np.asarray(pf)[:, None, :]) ** 2).sum(-1))   # :186-187
    time_index = np.zeros(N, dtype=int)
    for i in range(N):                                      # :188-193
        idx = np.where(dist_goal[i] >= 0.05)[0]
        time_index[i] = 0 if idx.size == 0 else idx[-1] + 2    # find(...,'last') is 1-based, then + 1
    return dict(violation=int(min_dist < rmin - 0.05), min_dist=float(min_dist), totdist=totdist, time_index=time_index,
                traj_time=float(time_index.max() * Ts), n_samples=len(t), p=p)


def postcheck(pk, vk, ak, pf, h, rmin, c, vmax=2.0, amax=1.0, Ts=0.01, pairs="literal"):
    """pk, vk, ak: [N, KT, 3] un-rescaled MPC histories of ONE scene (columns 1..k-1 of the .m arrays);
    pf: [N,3].  Returns dict(r_factor, h_scaled, violation, min_dist, totdist, traj_time, n_samples, p, pk)."""
    r_factor, _, _ = scale_factor(vk, ak, vmax, amax)
    h_scaled = h / np.sqrt(r_factor)                        # :146
    pk2, vk2, ak2 = rescale(pk, vk, ak, r_factor, h_scaled)
    out = interp_check(pk2, h_scaled, pf, rmin, c, Ts, pairs)
    out.update(r_factor=float(r_factor), h_scaled=float(h_scaled), pk=pk2, vk=vk2, ak=ak2)
    return out
