"""Solver-independent certificates for per-agent QP results (mandatory for the variants without a
recorded MATLAB/quadprog output: hard, ondemand, ellip, softall, repair, all3).

The literal dense QP  min 1/2 x'Hx + f'x  s.t.  Cx <= d,  x = [a; eps]  of an agent is taken from the oracle's ASSEMBLY
(oracle.assemble_one: the dense H, f, Ain, bin, lb, ub exactly as the .m files build them).  What is checked here uses NO
active-set / Goldfarb-Idnani code at all:

  * a reported solution `a`:  the slack part is completed in closed form (each eps_i appears in one row and two bounds),
    then  primal feasibility  max(Cx - d) <= tol_p,
          stationarity + dual feasibility + complementarity:  Lawson-Hanson NNLS (scipy.optimize.nnls) finds
          lam >= 0 supported on the rows with Cx - d >= -tol_act minimising |Hx + f + C_A' lam|; the residual must
          vanish (relative to the gradient scale).  Strict convexity then makes x THE minimiser.
  * a reported infeasibility: the phase-1 LP  min t  s.t.  Cx - t <= d, t >= 0  (scipy.optimize.linprog, HiGHS dual
    simplex / IPM) must have a strictly positive optimum.  For the retry-ladder variants the LAST ladder level is
    tested (the levels are nested: a larger |lb_eps| only relaxes the problem).
"""
import numpy as np
from scipy.optimize import linprog, nnls

N3 = 45


def complete_slack(qp, a):
    """x = [a; eps] with the optimal slack for this a: eps_i = min(0, (d_i - C_i a)/coef_i) clipped at its lower
    bound (the linear penalty is negative, so every slack wants to be as large as its row and eps <= 0 allow)."""
    C, d = qp["C"], qp["d"]
    n = C.shape[1]
    ns = n - N3
    x = np.zeros(n)
    x[:N3] = a
    if ns == 0:
        return x
    nc = qp["ncoll"]
    assert ns == nc
    res = C[:nc, :N3] @ a - d[:nc]                       # > 0: violated without slack
    coef = C[np.arange(nc), N3 + np.arange(nc)]
    eps = np.minimum(0.0, -res / coef)
    # lower bounds of the slack, if the variant has them: rows -eps_i <= -lb after the eps_i <= 0 rows
    m = C.shape[0]
    base = nc + 4 * N3
    if m >= base + 2 * ns:
        lb = -d[base + ns: base + 2 * ns]
        eps = np.maximum(eps, lb)
    x[N3:] = eps
    return x


def kkt_certificate(qp, a, tol_act=1e-7):
    """dict(primal, stat, stat_rel, lam_max, n_active) for the reported acceleration vector a."""
    H, f, C, d = qp["H"], qp["f"], qp["C"], qp["d"]
    x = complete_slack(qp, np.asarray(a, float))
    r = C @ x - d
    primal = float(r.max())
    g = H @ x + f
    scale_rows = np.maximum(1.0, np.abs(C).max(axis=1))
    act = np.where(r >= -tol_act * scale_rows)[0]
    if len(act):
        # column scaling keeps NNLS well conditioned when unit rows and position-space rows mix
        cn = np.linalg.norm(C[act], axis=1)
        A = (C[act] / cn[:, None]).T
        lam_s, resid = nnls(A, -g, maxiter=20 * max(A.shape))
        lam = lam_s / cn
        stat = float(np.abs(g + C[act].T @ lam).max())
    else:
        lam = np.zeros(0)
        stat = float(np.abs(g).max())
    gs = max(1.0, float(np.abs(g).max()))
    return dict(primal=primal, stat=stat, stat_rel=stat / gs, lam_max=float(lam.max()) if len(lam) else 0.0,
                n_active=int(len(act)), compl=float(np.abs(lam * r[act]).max()) if len(lam) else 0.0)


def lp_infeasibility(qp):
    """optimum t* of the phase-1 LP (min t : Cx - t <= d, t >= 0); > 0 <=> the constraint set is empty."""
    C, d = qp["C"], qp["d"]
    m, n = C.shape
    # equilibrate rows (position rows are O(h^2), box rows O(1)); t then measures the scaled violation
    s = 1.0 / np.maximum(1e-12, np.abs(C).max(axis=1))
    A = np.hstack([C * s[:, None], -np.ones((m, 1))])
    c = np.zeros(n + 1)
    c[-1] = 1.0
    res = linprog(c, A_ub=A, b_ub=d * s, bounds=[(None, None)] * n + [(0, None)], method="highs")
    assert res.status == 0, res.message
    return float(res.fun)
