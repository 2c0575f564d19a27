"""development (round 4): the longest agents of the 10^4-agent scene (C4) per MPC step: iterations, ladder tries, rows, largest working set"""
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
for k in range(9):
    o = d4.step_batch(l4, xp, xv, xa, pf4)
    inf = o["info"][0]; st = o["status"][0]
    it = inf[:, 4]
    top = np.argsort(it)[-8:][::-1]
    print(f"step {k + 2}: iters mean {it.mean():.1f}  >=64: {(it >= 64).sum()}  >=100: {(it >= 100).sum()}  tries>1: {(inf[:, 2] > 1).sum()}  status!=1: {(st != 1).sum()}")
    print("    top: " + "  ".join(f"[it {it[a]} tries {inf[a, 2]} rows {inf[a, 1]} maxq {inf[a, 7]} act {inf[a, 6]} st {st[a]}]" for a in top))
    ok = o["status"] == 1
    l4 = np.where(ok[..., None], o["p"], l4); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
