"""CPU ORACLE for the dense collision-row helpers (TEST INFRASTRUCTURE ONLY).

Literal numpy restatements (MATLAB array conventions: l is 3 x K x N_obs, n / k 1-based) of
  dec-iSCP/CollConstr.m:1-24, cup-SCP/AddCollConstr.m:1-31,
  dmpc/matlab/CollConstrSoftDMPC.m:1-32, CollConstrSoftDMPC2.m:1-32, CollConstrHardDMPC.m:1-36,
  CollConstrHardDMPCOnDemand.m, CollConstrEllipDMPC.m.
Parity unpinned: none of the reference's recorded workspaces stores an `Ain` of these helpers; the restatement
follows the .m text line by line (order = 2) and is cross-checked against the pinned solver oracle's rows in
tests/test_oracle_golden.py.
"""
import numpy as np


def _geom(p, pj, E1, E2, order=2):
    """dist = norm(E1*(p-pj),order); diff = (E2*(p-pj).^(order-1))'  (`.^` binds tighter than `*`); order 2 or 4"""
    dist = np.linalg.norm(E1 @ (p - pj), order)
    diff = E2 @ ((p - pj) ** (order - 1))
    return dist, diff


def CollConstrEllipDMPC_order(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order):
    """CollConstrEllipDMPC.m:1-30 literally, for ANY order (2 or 4: test/comp_test_ellipconstr.m:158-163 sets order = 4, E2 = E^-4):
    dist = norm(E1*(p-pj(:,k)),order); diff = (E2*(p-pj(:,k)).^(order-1))'; prev_dist = dist^(order-1);
    r = dist^(order-1)*(rmin - dist + diff*p/(dist^(order-1))) - diff*A_initp(3*(k-1)+1:3*k,:)*[po';vo']"""
    p, po, vo = (np.ravel(x).astype(float) for x in (p, po, vo))
    l = np.asarray(l, float)
    N_obs, K = l.shape[2], l.shape[1]
    Ain_total = np.zeros((N_obs - 1, 3 * K)); bin_total = np.zeros((N_obs - 1, 1)); prev_dist = np.zeros((N_obs - 1, 1))
    idx = 0
    for i in range(1, N_obs + 1):
        if i != n:
            pj = l[:, :, i - 1]
            d = p - pj[:, k - 1]
            dist = np.linalg.norm(E1 @ d, order)
            diff = E2 @ (d ** (order - 1))
            pd = dist ** (order - 1)
            r = pd * (rmin - dist + diff @ p / pd) - diff @ A_initp[3 * (k - 1):3 * k, :] @ np.r_[po, vo]
            diff_mat = np.r_[np.zeros(3 * (k - 1)), diff, np.zeros(3 * (K - k))]
            Ain_total[idx] = -diff_mat @ Ain
            bin_total[idx] = -r
            prev_dist[idx] = pd
            idx += 1
    return Ain_total, bin_total, prev_dist


def CollConstr(p, po, k, l, Ain, rmin, E1, E2, order=2):
    p, po = np.ravel(p).astype(float), np.ravel(po).astype(float)
    l = np.asarray(l, float)
    N_obs = l.shape[2] if l.size else 0
    Ain_total = np.zeros((N_obs, Ain.shape[1])); bin_total = np.zeros((N_obs, 1))
    for i in range(N_obs):
        pj = l[:, :, i]
        K = pj.shape[1]
        dist, diff = _geom(p, pj[:, k - 1], E1, E2, order)
        pd = dist ** (order - 1)
        r = pd * (rmin - dist + diff @ p / pd) - diff @ po
        diff_mat = np.r_[np.zeros(3 * (k - 2)), diff, np.zeros(3 * (K - (k - 1)))]
        Ain_total[i] = -diff_mat @ Ain
        bin_total[i] = -r
    return Ain_total, bin_total


def _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, sel_fn, nrows, k_ctr, dist_lt=None, order=2):
    p, po, vo = (np.ravel(x).astype(float) for x in (p, po, vo))
    l = np.asarray(l, float)
    N_obs, K = l.shape[2], l.shape[1]
    Ain_total = np.zeros((nrows, 3 * K)); bin_total = np.zeros((nrows, 1)); prev_dist = np.zeros((nrows, 1))
    idx = 0
    for i in range(1, N_obs + 1):
        if i != n and sel_fn(i):
            pj = l[:, :, i - 1]
            dist, diff = _geom(p, pj[:, k - 1], E1, E2, order)
            if dist_lt is not None and not dist < dist_lt:
                continue
            pd = dist ** (order - 1)
            r = pd * (rmin - dist + diff @ p / pd) - diff @ A_initp[3 * (k_ctr - 1):3 * k_ctr, :] @ np.r_[po, vo]
            diff_mat = np.r_[np.zeros(3 * (k_ctr - 1)), diff, np.zeros(3 * (K - k_ctr))]
            if idx >= Ain_total.shape[0]:
                Ain_total = np.vstack([Ain_total, np.zeros((1, 3 * K))]); bin_total = np.vstack([bin_total, [[0.0]]])
                prev_dist = np.vstack([prev_dist, [[0.0]]])
            Ain_total[idx] = -diff_mat @ Ain
            bin_total[idx] = -r
            prev_dist[idx] = pd
            idx += 1
    return Ain_total, bin_total, prev_dist


def CollConstrSoftDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    v = np.ravel(violation).astype(bool)
    return _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, lambda i: v[i - 1], int(v.sum()), k, order=order)


def CollConstrSoftDMPC2(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    v = np.ravel(violation).astype(bool)
    return _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, lambda i: v[i - 1], int(v.sum()), k - 1, order=order)


def CollConstrHardDMPCOnDemand(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    v = np.ravel(violation).astype(bool)
    return _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, lambda i: v[i - 1], int(v.sum()), k, order=order)[:2]


def CollConstrHardDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order):
    N_obs = np.asarray(l).shape[2]
    return _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, lambda i: True, N_obs - 1, k, dist_lt=1.0, order=order)[:2]


def CollConstrEllipDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order):
    N_obs = np.asarray(l).shape[2]
    return _dmpc(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, lambda i: True, N_obs - 1, k, order=order)


def AddCollConstr(p, po, K, rmin, A, E1, E2, order=2):
    p = np.asarray(p, float); po = np.asarray(po, float).reshape(-1, 3)
    N = p.shape[2]
    Ain_total = np.zeros((K * N * (N - 1) // 2, A.shape[1])); bin_total = np.zeros((K * N * (N - 1) // 2, 1))
    row = 0
    for i in range(N - 1):
        for j in range(i + 1, N):
            for k in range(K):
                d = p[:, k, i] - p[:, k, j]
                dist = np.linalg.norm(E1 @ d, order)
                diff = E2 @ (d ** (order - 1))
                r = dist ** (order - 1) * (rmin - dist) + diff @ d - diff @ (po[i] - po[j])
                diff_mat = np.zeros(3 * K * N)
                diff_mat[3 * K * i + 3 * k:3 * K * i + 3 * k + 3] = diff
                diff_mat[3 * K * j + 3 * k:3 * K * j + 3 * k + 3] = -diff
                Ain_total[row] = -diff_mat @ A
                bin_total[row] = -r
                row += 1
    return Ain_total, bin_total


# ---- dmpc/matlab helpers of the legacy SCP loop (solveDMPC.m) ---------------------------------------------------------------------
def CheckCollDMPC(p, l, n, k, r_min):
    """CheckCollDMPC.m:1-10 literally: violation = any_i~=n norm(p - l(:,k,i)) < r_min"""
    p = np.ravel(p).astype(float)
    l = np.asarray(l, float)
    violation = False
    for i in range(1, l.shape[2] + 1):
        if i != n:
            pj = l[:, :, i - 1]
            dist = np.linalg.norm(p - pj[:, k - 1])
            violation = violation or (dist < r_min)
    return violation


def CollConstrDMPC(p, po, vo, n, k, l, Ain, r_min, A_initp):
    """CollConstrDMPC.m:1-34 literally (one row per other agent at step k, linearised about p)."""
    p, po, vo = (np.ravel(x).astype(float) for x in (p, po, vo))
    l = np.asarray(l, float)
    rows, rhs = [], []
    if l.size:
        for i in range(1, l.shape[2] + 1):
            if i != n:
                pj = l[:, :, i - 1]
                K = pj.shape[1]
                dist = np.linalg.norm(p - pj[:, k - 1])
                diff = p - pj[:, k - 1]
                r = dist * (r_min - dist + (p - pj[:, k - 1]) @ p / dist) - (p - pj[:, k - 1]) @ A_initp[3 * (k - 1):3 * k, :] @ np.r_[po, vo]
                diff_mat = np.r_[np.zeros(3 * (k - 1)), diff, np.zeros(3 * (K - k))]
                rows.append(-diff_mat @ Ain)
                rhs.append(-r)
    return (np.array(rows).reshape(len(rows), np.asarray(Ain).shape[1]), np.array(rhs).reshape(-1, 1))


def maxDeviation(p, prev_p):
    """maxDeviation.m:1-11 literally: K = length(p)/3 (of the 3 x k_hor MATRIX: max(size)/3), dist(k) = norm(p(:,k) - prev_p(:,k)), max"""
    p, prev_p = np.asarray(p, float), np.asarray(prev_p, float)
    K = max(p.shape) / 3
    dist = [np.linalg.norm(p[:, k - 1] - prev_p[:, k - 1]) for k in range(1, int(K) + 1)]
    return max(dist)
