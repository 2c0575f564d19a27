"""development (round 4): how well does an agent's solve at MPC step k predict its solve at step k + 1 (the order hint)?  C4, N = 10^4,
closed loop: iterations per agent per step, rank correlation between consecutive steps, and where the heaviest agents of a step stood the step before"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg4, N4 = dict(wl.CONFIGS["C4"]), 10000
d4 = mp.Dmpc("bound", **wl.solver_kwargs(cfg4, N4))
po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)
l4, _, _ = d4.init_batch(po4, pf4)
xp, xv, xa = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
prev = None
for k in range(9):
    o = d4.step_batch(l4, xp, xv, xa, pf4)
    inf = o["info"][0]
    it = inf[:, 4].astype(float); rows = inf[:, 1].astype(float); tries = inf[:, 2]
    if prev is not None:
        pit, prow, _ = prev
        rk = lambda x: np.argsort(np.argsort(x))
        top = np.argsort(it)[-100:]
        print(f"step {k + 2}: rank corr(iters, prev iters) {np.corrcoef(rk(it), rk(pit))[0, 1]:.2f}; corr(iters, rows) {np.corrcoef(rk(it), rk(rows))[0, 1]:.2f}; "
              f"the 100 heaviest (>= {it[top].min():.0f} iterations): previous-step percentile median {np.median(rk(pit)[top]) / N4 * 100:.0f} / 10th {np.percentile(rk(pit)[top], 10) / N4 * 100:.0f}; "
              f"row-count percentile median {np.median(rk(rows)[top]) / N4 * 100:.0f} / 10th {np.percentile(rk(rows)[top], 10) / N4 * 100:.0f}; tries>1 among them {(tries[top] > 1).sum()}, had tries>1 before {(prev[2][top] > 1).sum()}")
    prev = (it, rows, tries.copy())
    ok = o["status"] == 1
    l4 = np.where(ok[..., None], o["p"], l4); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
