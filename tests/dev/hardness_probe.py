"""development probe (CPU, numpy model): which cheap per-agent features of the hard rows predict the active-set iteration
count (launch-order predictor for the persistent solve kernel)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import structured_model as sm   # tests/dev/structured_model.py (numpy model used by these probes only)
from oracle import oracle as orc
from multiagent_planning_amd import workload as wl

cfg = wl.CONFIGS["C2"]; N = 100; K = 15
kw = wl.solver_kwargs(cfg, N)
nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
po_all, pf_all = wl.make_scenes(cfg, nsc)
h = kw["h"]; alim = kw["alim"]
def pivot(c, v, Tm):
    if c.typ in (sm.BOXHI, sm.BOXLO): return v / np.sqrt(Tm.Hinv1[c.ka, c.ka])
    if c.typ in (sm.POSHI, sm.POSLO): return v / np.sqrt(Tm.P1[c.ky, c.ky])
    if c.typ == sm.COLL: return v / np.sqrt(Tm.P1[c.ky, c.ky] * (c.yv @ c.yv))
    return v * np.sqrt(2.0)
sm.PIVOT = pivot
feat = []
for s in range(nsc):
    po, pf = po_all[s], pf_all[s]
    l = np.stack([np.asarray(orc.init_one(po[i], pf[i], h, K)[0]).reshape(-1) for i in range(N)])
    for n in range(N):
        stats = []
        out = sm.solve_agent_model("hard", K, h, kw["rmin"], cfg["c"], alim, kw["Q1"], kw["S1"], -5e4, kw["pmin"], kw["pmax"], l, n, po[n], np.zeros(3), np.zeros(3), pf[n], stats)
        iters = sum(st["iters"] for st in stats) if stats else 0
        _, _, rows, _, _ = sm.scan_rows("hard", K, h, l, n, po[n], np.zeros(3), kw["rmin"], cfg["c"], -5e4)
        Tb = sm._TABLE_CACHE[(h, K, kw["Q1"], kw["S1"])]      # rows exist: collision cost case
        f = np.zeros((K, 3))
        for ax in range(3):
            f[:, ax] = -2.0 * Tb.q * Tb.L[K - 1] * (pf[n][ax] - po[n][ax])
        a_unc = -Tb.Hinv1 @ f
        w_unc = Tb.L @ a_unc
        m_min = 9.0; nviol = 0; vmax = 0.0; ntight = 0; mstep = {}
        for r in rows:
            hw = 0.5 * alim * ((r["kc"] + 1) * h) ** 2
            rng_ = hw * np.abs(r["xi"]).sum()
            m = (r["b"] + rng_) / (2 * rng_)          # 0: only the far corner of the box satisfies the row; 1: the whole box does
            m_min = min(m_min, m); ntight += m < 0.25
            v = -(r["xi"] @ w_unc[r["kc"]]) - r["b"]
            if v > 0: nviol += 1; vmax = max(vmax, v / np.sqrt(Tb.P1[r["kc"], r["kc"]] * (r["xi"] @ r["xi"])))
            mstep.setdefault(r["kc"], []).append(m)
        # per step: sum of (1 - m) of the two tightest rows (two conflicting tight rows cannot both hold)
        pair = max((sum(sorted(1 - np.array(v))[-2:]) for v in mstep.values()), default=0.0)
        nbox = int((np.abs(a_unc) > alim).sum())
        # rows violated at the box-clipped unconstrained minimiser
        w_clip = Tb.L @ np.clip(a_unc, -alim, alim)
        nviol_c = sum((-(r["xi"] @ w_clip[r["kc"]]) - r["b"]) > 0 for r in rows)
        steps_viol = len({r["kc"] for r in rows if (-(r["xi"] @ w_unc[r["kc"]]) - r["b"]) > 0})
        feat.append((iters, out["status"], len(rows), nviol, vmax, m_min, ntight, pair, nbox, nviol_c, steps_viol))
F = np.array(feat, float)
it = F[:, 0]
print("agents", len(F), "mean iters %.2f" % it.mean(), "infeasible", int((F[:, 1] == 8).sum()))
names = ["rows", "nviol", "vmax(Hnorm)", "-m_min", "ntight", "pair", "nbox", "nviol_clip", "steps_viol", "nviol+nbox", "-m_min*nviol", "nviol-10*m_min"]
cols = [F[:, 2], F[:, 3], F[:, 4], -F[:, 5], F[:, 6], F[:, 7], F[:, 8], F[:, 9], F[:, 10], F[:, 3] + F[:, 8], (1 - F[:, 5]) * (1 + F[:, 3]), F[:, 3] - 10 * F[:, 5]]
long_ = it > np.percentile(it, 90)
for nm, c in zip(names, cols):
    rk = np.argsort(-c)
    top = np.zeros(len(F), bool); top[rk[: len(F) // 5]] = True      # heaviest fifth by this feature
    print("%-12s corr %.2f   share of the 10%% longest agents in the top fifth: %.2f   share of all iterations there: %.2f" %
          (nm, np.corrcoef(c, it)[0, 1], (long_ & top).sum() / long_.sum(), it[top].sum() / it.sum()))
np.save("/tmp/hardness_feat.npy", F)
