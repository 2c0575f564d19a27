"""CPU: solver-independent certificates (tests/certificates.py) of the ORACLE's results for the variants that have no
recorded MATLAB/quadprog output (hard, ondemand, ellip, softall, repair, all3 -- parity unpinned at the solver
boundary).  Every solved agent must satisfy the KKT conditions of the literal dense QP (multipliers recovered by NNLS,
not by the oracle's active-set code) and every agent reported infeasible must have an empty constraint set by a phase-1
LP (HiGHS).  The pinned variants bound/bound2 run through the same check as a control."""
import numpy as np
import pytest

from oracle import oracle as orc
from helpers import load_golden, step14_inputs
import certificates as cert

UNPINNED = ["hard", "ondemand", "ellip", "softall", "repair", "all3", "cpp1", "softall_c", "scp"]


def check_batch(prm, l, xp, xv, xa, pf, a, status, tries, what, agents=None, tol=1e-8, t_min=1e-7):
    """certificate of every agent in `agents` (default all); returns (n_solved, n_infeasible, worst dict)."""
    worst = dict(primal=-1.0, stat_rel=0.0, compl_rel=0.0, t_min=np.inf)
    ns = ni = 0
    for n in (range(l.shape[0]) if agents is None else agents):
        st = int(status[n])
        if not (st & (orc.ST_SOLVED | orc.ST_INFEAS)):
            continue
        qp = orc.assemble_one(prm, l, n, xp[n], xv[n], xa[n], pf[n], level=max(int(tries[n]) - 1, 0))
        assert qp is not None, f"{what}: agent {n} has a result but the reference returns `coll`"
        if st & orc.ST_SOLVED:
            k = cert.kkt_certificate(qp, a[n])
            ns += 1
            worst["primal"] = max(worst["primal"], k["primal"])
            worst["stat_rel"] = max(worst["stat_rel"], k["stat_rel"])
            worst["compl_rel"] = max(worst["compl_rel"], k["compl"] / max(1.0, k["lam_max"]))
            assert k["primal"] <= tol, f"{what}: agent {n} primal infeasibility {k['primal']:.2e}"
            assert k["stat_rel"] <= tol, f"{what}: agent {n} stationarity residual {k['stat_rel']:.2e} (relative)"
            assert k["compl"] <= tol * max(1.0, k["lam_max"]), f"{what}: agent {n} complementarity {k['compl']:.2e}"
        else:
            t = cert.lp_infeasibility(qp)
            ni += 1
            worst["t_min"] = min(worst["t_min"], t)
            assert t > t_min, f"{what}: agent {n} reported infeasible but the phase-1 LP optimum is {t:.2e}"
    return ns, ni, worst


@pytest.mark.parametrize("variant", UNPINNED + ["bound", "bound2"])
def test_oracle_results_carry_certificates(variant):
    g, kw = load_golden("comp_kctr_3_bound2")   # N = 100, MPC step 14 of a recorded congested transition
    l, xp, xv, xa, pf = step14_inputs(g)
    prm = orc.make_params(variant, **kw)
    ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
    ns, ni, worst = check_batch(prm, l, xp, xv, xa, pf, ref["a"], ref["status"], ref["info"][:, orc.I_TRIES], f"oracle {variant}")
    assert ns > 30
    if variant in ("hard", "ondemand", "ellip"):
        assert ni > 0     # the slack-free variants do meet infeasible QPs on this scene
    print(f"KKT/LP certificate [{variant}] (parity unpinned at the solver boundary): {ns} solved, {ni} infeasible, worst {worst}")


def test_certificate_rejects_wrong_answers():
    """the checker itself: a perturbed solution and a feasible problem declared infeasible are both caught"""
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g)
    prm = orc.make_params("hard", **kw)
    ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
    n = int(np.where(ref["status"] == 1)[0][0])
    qp = orc.assemble_one(prm, l, n, xp[n], xv[n], xa[n], pf[n])
    good = cert.kkt_certificate(qp, ref["a"][n])
    a_bad = ref["a"][n].copy()
    a_bad[7] += 1e-5 if a_bad[7] < 0.9 else -1e-5
    bad = cert.kkt_certificate(qp, a_bad)
    assert good["stat_rel"] < 1e-10 and max(bad["stat_rel"], bad["primal"]) > 1e-7
    assert cert.lp_infeasibility(qp) < 1e-12            # feasible problem: optimum 0
    m = int(np.where(ref["status"] == 8)[0][0])
    assert cert.lp_infeasibility(orc.assemble_one(prm, l, m, xp[m], xv[m], xa[m], pf[m])) > 1e-6
