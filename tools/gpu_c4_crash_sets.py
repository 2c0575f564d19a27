"""development: which acceleration bounds are violated at the unconstrained minimiser of the agents of the 10^4-agent scene (C4, MPC steps 2-4)?
Per axis: a prefix of the horizon (the crash start copies its factor columns from the table), a prefix plus a run at the END of the
horizon, or something else -- the shapes the product rounds of crash_append are left with."""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
N = 10000; K = 15
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc(cfg["variant"], **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
Lam, Av, A0, Dl = mp.model_matrices(kw["h"])
L1 = Lam[::3, ::3]; D1 = Dl[::3, ::3]          # per-axis 15x15 blocks
h = kw["h"]; alim = kw["alim"]
for step in range(1, 5):
    out = d.step_batch(l, xp, xv, xa, pf)
    inf = out["info"][0]
    # cost case of every agent as the solver recorded it (0 far, 1 near, 2 rows); weights as in upload_tables
    qs = np.array([1000.0, 10000.0, kw["Q1"]]); ss = np.array([10.0, 10.0, kw["S1"]])
    cc = inf[:, 3]
    cnt = collections.Counter(); nb_tab = 0; nb_rest = 0; nag = 0
    for c in range(3):
        idx = np.where(cc == c)[0]
        if not len(idx): continue
        H = 2 * (qs[c] * np.outer(L1[K - 1], L1[K - 1]) + ss[c] * D1.T @ D1 + np.eye(K)); Hi = np.linalg.inv(H)
        for ax in range(3):
            g = pf[0, idx, ax] - (xp[0, idx, ax] + K * h * xv[0, idx, ax])
            f = -2 * (qs[c] * np.outer(g, L1[K - 1]))
            f[:, 0] -= 2 * ss[c] * xa[0, idx, ax]
            au = -(f @ Hi)                                   # [agents][K]
            V = np.abs(au) > alim + 1e-10
            for v in V:
                n = int(v.sum())
                if n == 0: cnt["none"] += 1; continue
                m = 0
                while m < K and v[m]: m += 1
                rest = v[m:]
                nb_tab += m; nb_rest += int(rest.sum())
                if not rest.any(): cnt["prefix"] += 1
                else:
                    r = np.where(rest)[0] + m
                    contiguous = (r[-1] - r[0] + 1) == len(r)
                    if m > 0 and contiguous and r[-1] == K - 1: cnt["prefix + end run"] += 1
                    elif m == 0 and contiguous and r[-1] == K - 1: cnt["end run only"] += 1
                    elif m == 0 and contiguous: cnt[f"middle run from step {r[0]}"] += 1
                    else: cnt["other"] += 1
        nag += len(idx)
    print(f"MPC step {step}: agents {nag}; violated bounds per agent from the prefix {nb_tab/nag:.1f}, outside it {nb_rest/nag:.1f}; axis shapes: " +
          ", ".join(f"{k} {v}" for k, v in cnt.most_common(8)))
    ok = (out["status"] == 1)[..., None]
    l = np.where(ok, out["p"], l); xp = np.where(ok, out["p"][..., :3], xp); xv = np.where(ok, out["v"][..., :3], xv); xa = np.where(ok, out["a"][..., :3], xa)
