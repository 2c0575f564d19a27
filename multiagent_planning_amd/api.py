"""Host-side mirror of the reference's MATLAB interface for the DMPC hot path.

Same function names, positional argument lists, return tuples and failure conventions as the
`.m` files under dmpc/matlab/ (SURVEY.md section 8b), so code (and tests) written against the reference
reads the same.  Every solver call goes through the C ABI of libdmpc_hip.so to the HIP kernel; there
is no CPU path here -- without the library or a GPU the calls raise DmpcError.

Array conventions are MATLAB's: `l` is 3 x K x N, states are length-3 vectors, `n` is 1-based,
outputs p, v, a are 3 x K; on failure p = v = a = empty (0 x 0) exactly as the reference returns `[]`.
"""
import numpy as np

from . import _lib
from ._lib import Dmpc, ST_COLL, ST_INFEAS, ST_OUTBOUND, ST_SOLVED, ST_CAPACITY, ST_ITERCAP

_EMPTY = np.zeros((0, 0))
_ctx_cache = {}


def _table(l):
    """MATLAB l(3,K,N) -> rows [N, 3K] (same memory order as column-major MATLAB storage)."""
    l = np.asarray(l, dtype=np.float64)
    if l.ndim != 3 or l.shape[0] != 3:
        raise ValueError("l must be 3 x K x N")
    return np.ascontiguousarray(l.transpose(2, 1, 0).reshape(l.shape[2], -1))


def _ctx(variant, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term, device=0, tol=0.0):
    if int(K) != _lib.K_HOR:
        raise ValueError("K (k_hor) must be 15")
    c = 1.0 / float(np.asarray(E1)[2, 2])
    key = (variant, float(h), float(rmin), tuple(map(float, np.ravel(pmin))), tuple(map(float, np.ravel(pmax))),
           float(alim), float(Q1), float(S1), c, int(order), float(term), device, float(tol))
    d = _ctx_cache.get(key)
    if d is None:
        d = Dmpc(variant, device=device, h=h, rmin=rmin, c=c, alim=alim, Q1=Q1, S1=S1, term=term,
                 pmin=tuple(np.ravel(pmin)), pmax=tuple(np.ravel(pmax)), order=order, tol=tol)
        _ctx_cache[key] = d
    return d


def _solve(variant, po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term):
    d = _ctx(variant, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term)
    r = d.solve_one(_table(l), int(n) - 1, np.ravel(po), np.ravel(vo), np.ravel(ao), np.ravel(pf))
    st = r["status"]
    if st & (ST_CAPACITY | ST_ITERCAP):
        raise _lib.DmpcError(f"internal capacity/iteration limit hit (status {st}); result not valid")
    if st & ST_SOLVED:
        p, v, a = (r[k].reshape(K, 3).T.copy() for k in ("p", "v", "a"))   # vec2mat(.,3)'
    else:
        p = v = a = _EMPTY
    return p, v, a, st


# ---- solver entry points (signatures of the .m files) ------------------------------------------

def solveSoftDMPCbound(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, A_p, A_v, Delta, Q1, S1, E1, E2, order, term):
    """[p,v,a,feasible,outbound,coll] = solveSoftDMPCbound(...)  (solveSoftDMPCbound.m:1).
    coll -> feasible = 1 with empty outputs (:25-31); outbound keeps feasible = 1 (:125-128)."""
    p, v, a, st = _solve("bound", po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term)
    coll = int(bool(st & ST_COLL))
    feasible = int(bool(st & (ST_SOLVED | ST_COLL)))
    return p, v, a, feasible, int(bool(st & ST_OUTBOUND)), coll


def _success6(variant, args, term):
    (po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, A_p, A_v, Delta, Q1, S1, E1, E2, order) = args
    p, v, a, st = _solve(variant, po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term)
    outbound = int(bool(st & ST_OUTBOUND))
    success = int(bool(st & ST_SOLVED) and not outbound)   # solveSoftDMPCbound2.m:123-126: outbound -> success = 0
    return p, v, a, success, outbound, int(bool(st & ST_COLL))


def solveSoftDMPCbound2(*args):
    """[p,v,a,success,outbound,coll] = solveSoftDMPCbound2(...23 args...)  (solveSoftDMPCbound2.m:1)."""
    return _success6("bound2", args[:22], args[22])


def solveSoftDMPCall(*args):
    """[p,v,a,success,outbound,coll] = solveSoftDMPCall(...23 args...)  (solveSoftDMPCall.m:1)."""
    return _success6("all3", args[:22], args[22])


def solveHardDMPC(*args):
    """[p,v,a,success,outbound,coll] = solveHardDMPC(...22 args, no term...)  (solveHardDMPC.m:1)."""
    return _success6("hard", args[:22], -5e4)


def solveHardDMPCOnDemand(*args):
    """[p,v,a,success,outbound,coll] = solveHardDMPCOnDemand(...22 args...)  (solveHardDMPCOnDemand.m:1)."""
    return _success6("ondemand", args[:22], -5e4)


def solveSoftDMPCrepair(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, Delta, Q1, S1, E1, E2, order, term):
    """[p,v,a,success,outbound,coll] = solveSoftDMPCrepair(...)  (solveSoftDMPCrepair.m:1)."""
    p, v, a, st = _solve("repair", po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term)
    outbound = int(bool(st & ST_OUTBOUND))
    return p, v, a, int(bool(st & ST_SOLVED) and not outbound), outbound, int(bool(st & ST_COLL))


def solveSoftDMPC(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, Delta, Q1, S1, E1, E2, order):
    """[p,v,a,success,outbound] = solveSoftDMPC(...)  (solveSoftDMPC.m:1; failure without a
    violation is reported as outbound, :88-96)."""
    d = _ctx("softall", h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -1e5)
    r = d.solve_one(_table(l), int(n) - 1, np.ravel(po), np.ravel(vo), np.ravel(ao), np.ravel(pf))
    st = r["status"]
    if st & ST_SOLVED:
        return tuple(r[k].reshape(K, 3).T.copy() for k in ("p", "v", "a")) + (1, 0)
    return _EMPTY, _EMPTY, _EMPTY, 0, int(r["info"][_lib.I_VIOLK] == 0)


def solveSoftDMPC_c(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, Delta, Q1, S1, E1, E2, order):
    """[p,v,a,success] = solveSoftDMPC_c(...)  (solveSoftDMPC_c.m:1; test/comp_confidence.m:184): solveSoftDMPC with the slack penalties
    -1e4 (K/k)^2 and 1e6 (K/k)^2 (:60-63); `isempty(x)` -> p = v = a = [], success = 0 (:80-87), else success = exitflag = 1."""
    d = _ctx("softall_c", h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -1e4)
    r = d.solve_one(_table(l), int(n) - 1, np.ravel(po), np.ravel(vo), np.ravel(ao), np.ravel(pf))
    st = r["status"]
    if st & (ST_CAPACITY | ST_ITERCAP):
        raise _lib.DmpcError(f"internal capacity/iteration limit hit (status {st}); result not valid")
    if st & ST_SOLVED:
        return tuple(r[k].reshape(K, 3).T.copy() for k in ("p", "v", "a")) + (1,)
    return _EMPTY, _EMPTY, _EMPTY, 0


def solveDMPC(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, Delta, tol, Q1, S1):
    """[p,v,a,success] = solveDMPC(po,pf,vo,ao,n,h,l,K,rmin,pmin,pmax,alim,A,A_initp,Delta,tol,Q1,S1)  (solveDMPC.m:1; dmpc/matlab/dmpc.m:79):
    the legacy SCP loop -- up to k_hor passes of spherical hard rows re-linearised about the previous pass -- all of it inside one kernel
    launch (DMPC_VAR_SCP).  An infeasible pass returns p = v = [] and success = 0 (:58-63; `a` is quadprog's empty result)."""
    d = _ctx("scp", h, K, rmin, pmin, pmax, alim, Q1, S1, np.eye(3), 2, -5e4, tol=tol)
    r = d.solve_one(_table(l), int(n) - 1, np.ravel(po), np.ravel(vo), np.ravel(ao), np.ravel(pf))
    st = r["status"]
    if st & (ST_CAPACITY | ST_ITERCAP):
        raise _lib.DmpcError(f"internal capacity/iteration limit hit (status {st}); result not valid")
    if st & ST_SOLVED:
        return tuple(r[k].reshape(K, 3).T.copy() for k in ("p", "v", "a")) + (1,)
    return _EMPTY, _EMPTY, _EMPTY, 0


def maxDeviation(p, prev_p):
    """tol = maxDeviation(p, prev_p)  (maxDeviation.m:1-11); p, prev_p are 3 x K.  Restated as written: `K = length(p)/3` of the matrix is
    max(3, K)/3, so only the first 5 of 15 horizon steps are looked at."""
    return _rowctx().max_deviation(np.asarray(p, float).T, np.asarray(prev_p, float).T)


def CheckCollDMPC(p, l, n, k, r_min):
    """violation = CheckCollDMPC(p,l,n,k,r_min)  (CheckCollDMPC.m:1-10): any other agent closer than r_min (plain Euclidean norm) at step k."""
    l = np.asarray(l, float)
    N = l.shape[2]
    others = [j for j in range(N) if j != n - 1]
    if not others:
        return False
    _, _, dist = _rowctx().coll_rows(_obst(l), others, k - 1, 0, np.ravel(p), np.zeros(3), r_min, 1.0, np.zeros((3, 1)))
    return bool((dist < r_min).any())


def CollConstrDMPC(p, po, vo, n, k, l, Ain, r_min, A_initp):
    """[Ain_total, bin_total] = CollConstrDMPC(p,po,vo,n,k,l,Ain,r_min,A_initp)  (CollConstrDMPC.m:1-34): one spherical row per other agent at
    horizon step k, linearised about p."""
    l = np.asarray(l, float)
    Ain = np.asarray(Ain, float)
    if l.size == 0:
        return np.zeros((0, Ain.shape[1])), np.zeros((0, 1))
    N = l.shape[2]
    sel = [j for j in range(N) if j != n - 1]
    if not sel:
        return np.zeros((0, Ain.shape[1])), np.zeros((0, 1))
    A, b, _ = _dmpc_rows(p, po, vo, n, k, l, r_min, Ain, A_initp, np.eye(3), np.eye(3), 2, sel, k)
    return A, b[:, None]


def solveEllipDMPC(po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, A, A_initp, Delta, Q1, S1, E1, E2, order):
    """[p,v,a,success,outbound] = solveEllipDMPC(...)  (solveEllipDMPC.m:1)."""
    p, v, a, st = _solve("ellip", po, pf, vo, ao, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, -5e4)
    return p, v, a, int(bool(st & ST_SOLVED)), 0


# ---- model / helper functions -------------------------------------------------------------------

def getPosMat(h, K):
    """Apos = getPosMat(h,K)  (getPosMat.m:1): acceleration -> position matrix, 3K x 3K."""
    return _lib.model_matrices(h, K)[0]


def getPosVelMat(h, K):
    """Aaug = getPosVelMat(h,K)  (getPosVelMat.m:1): [x_K (6 rows); a_K (3); a_1 (3)] selectors, 12 x 3K."""
    return _lib.posvel_matrix(h, K)


def getDeltaMat(k_hor):
    """Delta = getDeltaMat(k_hor)  (getDeltaMat.m:1)."""
    return _lib.model_matrices(0.2, k_hor)[3]


def getModelMats(h, k_hor):
    """(A_p, A_v, A_initp) of the precompute loop dmpc_soft_bound.m:92-108."""
    Lam, Av, A0, _ = _lib.model_matrices(h, k_hor)
    return Lam, Av, A0


def propStatedmpc(po, vo, a, A_initp, A_p, A_v):
    """[p,v] = propStatedmpc(po,vo,a,A_initp,A_p,A_v)  (propStatedmpc.m:1-8); stacked 3K vectors.
    (Inside the solvers this step is fused into the HIP kernel; the standalone form runs the same two products on
    the device through dmpc_prop_state.)"""
    return _rowctx().prop_state(A_p, A_v, a, A_initp=A_initp, po=po, vo=vo, off_v=vo)


def propState(po, a, A_p, A_v, K):
    """[p,v] = propState(po,a,A_p,A_v,K)  (dec-iSCP/propState.m:1-10): zero initial velocity, the
    initial state prepended."""
    po = np.ravel(po).astype(float)
    new_p, new_v = _rowctx().prop_state(A_p, A_v, a, off_p=po)           # new_p + repmat(po', K-1, 1)
    return np.r_[po, new_p], np.r_[np.zeros(3), new_v]


def initDMPC(po, pf, h, k_hor, K):
    """[p,v,a] = initDMPC(po,pf,h,k_hor,K)  (initDMPC.m:1-13), 3 x k_hor each (dmpc_init_batch)."""
    if int(k_hor) != 15:
        raise NotImplementedError("horizon k_hor = 15 only (SURVEY.md section 8)")
    d = _ctx("bound", h, k_hor, 0.35, (0, 0, 0), (1, 1, 1), 1.0, 1000.0, 100.0, np.diag([1.0, 1.0, 0.5]), 2, -5e4)
    p, v, a = d.init_batch(np.ravel(po).astype(float)[None, None, :], np.ravel(pf).astype(float)[None, None, :])
    f = lambda x: np.ascontiguousarray(np.asarray(x).reshape(k_hor, 3).T)
    return f(p), f(v), f(a)


def is_inbounds(p, pmin, pmax):
    """inbounds = is_inbounds(p,pmin,pmax)  (is_inbounds.m:1-6); p is 3 x n."""
    return _rowctx().is_inbounds(np.asarray(p, float).reshape(3, -1).T, pmin, pmax)


def ReachedGoal(p, pf, length_t, error_tol, N):
    """pass = ReachedGoal(p,pf,length_t,error_tol,N)  (ReachedGoal.m:1-11); p is 3 x T x N, pf 1 x 3 x N."""
    p = np.asarray(p, float)
    if N > 1:
        cur, goal = p[:, length_t - 1, :].T, np.asarray(pf, float).reshape(3, N).T
    else:
        cur, goal = p[:, length_t - 1].reshape(1, 3), np.ravel(pf).reshape(1, 3)
    return _rowctx().reached_goal(cur, goal, error_tol)


# ---- a5/a6 helpers -------------------------------------------------------------------------------

def CheckCollSoftDMPC(p, l, n, k, E1, rmin, order):
    """[violation,min_dist,viol_constr] = CheckCollSoftDMPC(p,l,n,k,E1,rmin,order)  (CheckCollSoftDMPC.m:1-17).
    One scan step on its own (inside the solvers the whole scan runs fused in dmpc_scan_kernel): the ellipsoidal
    distances come from the device row builder (dmpc_coll_rows), the two thresholds are applied to them here."""
    l = np.asarray(l, float)
    N = l.shape[2]
    E1 = np.asarray(E1, float)
    c = _c_of(E1, np.linalg.matrix_power(E1, int(order)), order)
    others = [j for j in range(N) if j != n - 1]
    d = np.full(N, np.inf)
    if others:
        _, _, dist = _rowctx(order).coll_rows(_obst(l), others, k - 1, 0, np.ravel(p), np.zeros(3), rmin, c, np.zeros((3, 1)))
        d[others] = dist if int(order) == 2 else np.cbrt(dist)      # (the builder returns prev_dist = dist^(order-1))
    violation = (d < rmin).astype(float)
    viol_constr = (d < rmin * 3).astype(float)
    return violation, float(d.min()) if others else np.inf, viol_constr


def collision_rows(variant, po, vo, n, h, l, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term=-5e4):
    """Scan + collision rows of agent n exactly as solver `variant` builds them, computed by the GPU scan
    kernel (dmpc_rows_one) and expanded to the reference's dense form:
    returns (Ain [Nv x 3K], bin [Nv], prev_dist [Nv], k_viol, coll) -- cf. CollConstrSoftDMPC.m:1 and variants."""
    d = _ctx(variant, h, K, rmin, pmin, pmax, alim, Q1, S1, E1, order, term)
    r = d.rows_one(_table(l), int(n) - 1, np.ravel(po), np.ravel(vo))
    Lam = _lib.model_matrices(h, K)[0]
    nr = len(r["kc"])
    Ain = _rowctx().rows_dense(r["xi"], r["kc"], Lam)   # -diff_mat*Ain (CollConstrSoftDMPC.m:27) on the device
    if variant in ("bound", "bound2", "all3", "repair", "cpp", "cpp2"):
        dist = r["slack_coef"].copy()                 # the slack column is diag(prev_dist) in these variants
    else:                                              # recovered from xi = E2 (p - p_j): |E1 (p - p_j)|
        c = 1.0 / float(np.asarray(E1)[2, 2])
        dist = np.sqrt(r["xi"][:, 0] ** 2 + r["xi"][:, 1] ** 2 + (r["xi"][:, 2] * c) ** 2)
    return Ain, r["rhs"].copy(), dist, r["viol_k"], int(bool(r["status"] & ST_COLL))


# ---------------------------------------------------------------------------------------------
# CollConstr* / AddCollConstr: the dense collision rows the reference's helpers return, built on the GPU
# ---------------------------------------------------------------------------------------------
def _c_of(E1, E2, order):
    """the c of E = diag(1,1,c) behind E1 = E^-1 and E2 = E^-order (dmpc_soft_bound.m:20-22; order 4: test/comp_test_ellipconstr.m:160-163)"""
    E1, E2 = np.asarray(E1, float), np.asarray(E2, float)
    if int(order) not in (2, 4):
        raise NotImplementedError("ellipsoid order must be 2 or 4 (the values the reference's scripts use)")
    if not (np.allclose(E1, np.diag(np.diag(E1))) and abs(E1[0, 0] - 1) < 1e-15 and abs(E1[1, 1] - 1) < 1e-15
            and np.allclose(E2, np.linalg.matrix_power(E1, int(order)), rtol=0, atol=1e-15)):
        raise NotImplementedError("E1 must be diag(1,1,1/c) and E2 = E1^order")
    return 1.0 / E1[2, 2]


def _obst(l):
    """MATLAB l(3,K,N_obs) -> [N_obs,K,3]."""
    l = np.asarray(l, float)
    return np.ascontiguousarray(l.transpose(2, 1, 0)) if l.size else np.zeros((0, 1, 3))


_ROWCTX = {}


def _rowctx(order=2, device=0):
    """the context behind the standalone helpers; the dense row builders take the ellipsoid order from it (dmpc_params.order: an order-4
    context is one of an all-neighbour variant)"""
    key = (int(order), device)
    if key not in _ROWCTX:
        _ROWCTX[key] = _lib.Dmpc("bound", device=device) if int(order) == 2 else _lib.Dmpc("ellip", device=device, order=int(order))
    return _ROWCTX[key]


def CollConstr(p, po, k, l, Ain, rmin, E1, E2, order):
    """[Ain_total, bin_total] = CollConstr(p,po,k,l,Ain,rmin,E1,E2,order)   (dec-iSCP/CollConstr.m:1-24):
    rows of time step k (1-based, >= 2) against every obstacle in l(3,K,N_obs); the non-zero block of diff_mat is
    block k-1 (`zeros(1,3*(k-2))`, :17) and the rhs uses the agent's own start po (:14)."""
    lo = _obst(l)
    N_obs = lo.shape[0] if np.asarray(l).size else 0
    Ain = np.asarray(Ain, float)
    if N_obs == 0:
        return np.zeros((0, Ain.shape[1])), np.zeros((0, 1))
    c = _c_of(E1, E2, order)
    A, b, _ = _rowctx(order).coll_rows(lo, np.arange(N_obs), k - 1, k - 2, np.ravel(p), np.ravel(po), rmin, c, Ain)
    return A, b[:, None]


def _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k_ctr):
    lo = _obst(l)
    c = _c_of(E1, E2, order)
    # A_initp(3*(k_ctr-1)+1:3*k_ctr,:)*[po';vo'] (CollConstrSoftDMPC.m:21), on the device like the rest of the row
    a0, _ = _rowctx().prop_state(np.zeros((3, 1)), np.zeros((3, 1)), [0.0], A_initp=np.asarray(A_initp, float)[3 * (k_ctr - 1):3 * k_ctr, :],
                                 po=po, vo=vo)
    return _rowctx(order).coll_rows(lo, sel, k - 1, k_ctr - 1, np.ravel(p), a0, rmin, c, np.asarray(Ain, float))


def CollConstrSoftDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    """[Ain_total,bin_total,prev_dist] = CollConstrSoftDMPC(...)   (dmpc/matlab/CollConstrSoftDMPC.m:1-32)."""
    v = np.ravel(violation).astype(bool)
    sel = [i for i in range(v.size) if i != n - 1 and v[i]]
    A, b, d = _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k)
    return _pad(A, b, d, int(v.sum()))


def CollConstrSoftDMPC2(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    """same with the constraint enforced one step earlier, k_ctr = k-1   (CollConstrSoftDMPC2.m:8)."""
    v = np.ravel(violation).astype(bool)
    sel = [i for i in range(v.size) if i != n - 1 and v[i]]
    A, b, d = _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k - 1)
    return _pad(A, b, d, int(v.sum()))


def CollConstrHardDMPCOnDemand(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, violation):
    """[Ain_total,bin_total] = CollConstrHardDMPCOnDemand(...)   (CollConstrHardDMPCOnDemand.m)."""
    v = np.ravel(violation).astype(bool)
    sel = [i for i in range(v.size) if i != n - 1 and v[i]]
    A, b, d = _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k)
    return _pad(A, b, d, int(v.sum()))[:2]


def CollConstrHardDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order):
    """[Ain_total,bin_total] = CollConstrHardDMPC(...)   (CollConstrHardDMPC.m:1-36): every j != n closer than 1 (:19);
    the preallocated N_obs-1 rows that stay unused remain zero, as in the reference."""
    N_obs = np.asarray(l).shape[2]
    sel = [i for i in range(N_obs) if i != n - 1]
    A, b, d = _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k)
    keep = d < 1
    return _pad(A[keep], b[keep], d[keep], N_obs - 1)[:2]


def CollConstrEllipDMPC(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order):
    """[Ain_total,bin_total,prev_dist] = CollConstrEllipDMPC(...)   (CollConstrEllipDMPC.m): all j != n."""
    N_obs = np.asarray(l).shape[2]
    sel = [i for i in range(N_obs) if i != n - 1]
    A, b, d = _dmpc_rows(p, po, vo, n, k, l, rmin, Ain, A_initp, E1, E2, order, sel, k)
    return A, b[:, None], d[:, None]


def _pad(A, b, d, nrows):
    """the reference preallocates `N_violation` rows (zeros) and fills the first idx-1 of them."""
    Ao, bo, do = np.zeros((nrows, A.shape[1])), np.zeros((nrows, 1)), np.zeros((nrows, 1))
    m = min(nrows, A.shape[0])
    Ao[:m], bo[:m, 0], do[:m, 0] = A[:m], b[:m], d[:m]
    if A.shape[0] > nrows:   # MATLAB grows the arrays when more rows are written than preallocated
        Ao, bo, do = A, b[:, None], d[:, None]
    return Ao, bo, do


def AddCollConstr(p, po, K, rmin, A, E1, E2, order):
    """[Ain_total, bin_total] = AddCollConstr(p,po,K,rmin,A,E1,E2,order)   (cup-SCP/AddCollConstr.m:1-31):
    p(3,K,N) previous trajectories, po(1,3,N); K N(N-1)/2 pairwise rows of the coupled QP over [a_1; ...; a_N]."""
    p = np.asarray(p, float)
    N = p.shape[2]
    c = _c_of(E1, E2, order)
    A = np.asarray(A, float)
    if N < 2:
        return np.zeros((0, A.shape[1])), np.zeros((0, 1))
    po = np.asarray(po, float)
    po = np.ascontiguousarray(po[0].T) if po.ndim == 3 else po.reshape(N, 3)      # MATLAB po(1,3,N) or [N,3]
    Ain, b = _rowctx(order).add_coll_constr(np.ascontiguousarray(p.transpose(2, 1, 0)), po, rmin, c, A)
    return Ain, b[:, None]
