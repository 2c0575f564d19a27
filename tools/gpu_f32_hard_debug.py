"""development (round 4): the agent(s) of the C2 hard sweep whose fp32-factor result differs from fp64 by more than 1e-9"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from helpers import init_table
cfg = wl.CONFIGS["C2"]; N, S = 100, 8; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 5)
d64, d32 = mp.Dmpc("hard", **kw), mp.Dmpc("hard", precision="f32factor", **kw)
l = np.stack([init_table(po[s], pf[s]) for s in range(S)])
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
Lam, Av, A0, Dl = mp.model_matrices(kw["h"])
for k in range(8):
    a = d64.step_batch(l, xp, xv, xa, pf); b = d32.step_batch(l, xp, xv, xa, pf)
    ok = (a["status"] == 1) & (b["status"] == 1)
    e = np.abs(a["p"] - b["p"]).max(-1) * ok
    for s, n in zip(*np.where(e > 1e-9)):
        print(f"step {k+2} scene {s} agent {n}: |dp| {e[s,n]:.2e}  fp64 info {a['info'][s,n]}  fp32 info {b['info'][s,n]}  |da| {np.abs(a['a'][s,n]-b['a'][s,n]).max():.2e}")
        for nm, o in (("fp64", a), ("f32T", b)):
            acc = o["a"][s, n]
            # cost 1/2 a'Ha + f'a needs H: use tracking cost directly: q|p_K - pf|^2 + s|Delta a - ..|^2 + |a|^2 (case 2: Q1, S1)
            pK = o["p"][s, n][-3:]
            da = np.diff(np.concatenate([xa[s, n][None], acc.reshape(15, 3)]), axis=0)
            cost = kw["Q1"] * ((pK - pf[s, n]) ** 2).sum() + kw["S1"] * (da ** 2).sum() + (acc ** 2).sum()
            print(f"    {nm}: cost {cost:.12f}  max|a| {np.abs(acc).max():.12f}")
    upd = (a["status"] & 1) == 1
    l = np.where(upd[..., None], a["p"], l); xp = np.where(upd[..., None], a["p"][..., :3], xp)
    xv = np.where(upd[..., None], a["v"][..., :3], xv); xa = np.where(upd[..., None], a["a"][..., :3], xa)
