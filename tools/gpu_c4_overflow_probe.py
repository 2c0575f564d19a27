"""development: C4 (ONE scene of 10^4 agents, solveSoftDMPCbound): which agents outgrow the 48-slot first tier, and can the scan see them coming?
For MPC steps 3-10 (teacher-forced): the largest working set of every agent (single 64-slot tier) against the number of acceleration bounds
violated at the unconstrained minimiser (what the crash start appends) and the agent's row count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg, N = dict(wl.CONFIGS["C4"]), 10000
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
po, pf = po[0], pf[0]
d = mp.Dmpc("bound", **kw).debug_option("tier1_qcap", 64)
Lam, Av, A0, Dl = mp.model_matrices(kw["h"])
L1 = Lam[0::3, 0::3]; D1 = Dl[0::3, 0::3]; K = 15; h = kw["h"]
def a_unc(xp, xv, xa, pf, rows):
    dn = np.linalg.norm(xp - pf, axis=1)
    q = np.where(rows, 1000.0, np.where(dn >= 1.0, 1000.0, 10000.0)); s = np.where(rows, kw["S1"], 10.0)
    out = np.zeros((len(xp), K, 3))
    for qq, ss in set(zip(q.tolist(), s.tolist())):
        m = (q == qq) & (s == ss)
        H1 = 2 * (qq * np.outer(L1[-1], L1[-1]) + ss * D1.T @ D1 + np.eye(K))
        g = pf[m] - (xp[m] + K * h * xv[m])                               # [n,3]
        f = -2 * (qq * L1[-1][None, :, None] * g[:, None, :]); f[:, 0, :] -= 2 * ss * xa[m]
        out[m] = -np.einsum("ij,njk->nik", np.linalg.inv(H1), f)
    return out
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(9):
    o = d.step_batch(l, xp, xv, xa, pf)
    inf = o["info"]; maxq = inf[:, 7]; rows = inf[:, 1]
    au = a_unc(xp, xv, xa, pf, rows > 0)
    nb = (np.abs(au) > kw["alim"] + 1e-10).reshape(N, -1).sum(1)
    over = maxq >= 48
    if k >= 1:
        pred = {}
        for name, score in (("nb", nb), ("nb+rows", nb + rows), ("nb+2rows", nb + 2 * rows)):
            for thr in (36, 40, 44, 48):
                p = score >= thr
                pred[f"{name}>={thr}"] = (int((p & over).sum()), int(p.sum()))
        print(f"step {k+2}: maxq histogram >=44: {dict(zip(*np.unique(maxq[maxq >= 44], return_counts=True)))}")
        print(f"step {k+2}: overflow agents {int(over.sum())}, max iters among them {int(inf[over, 4].max()) if over.any() else 0}; mean nb {nb.mean():.1f} rows {rows.mean():.1f}; "
              f"(caught, flagged) by predictor: {pred}")
    ok = o["status"] == 1
    l = np.where(ok[:, None], o["p"], l); xp = np.where(ok[:, None], o["p"][:, :3], xp)
    xv = np.where(ok[:, None], o["v"][:, :3], xv); xa = np.where(ok[:, None], o["a"][:, :3], xa)
