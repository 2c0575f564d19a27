"""development: which agents make the heavy MPC steps of ONE 100-agent scene (closed loop through step_batch; per-step
maximum iteration count, its agent's tries / active set / rows, and the step's total)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, 100)
po, pf = wl.make_scenes(cfg, 1, 100, wl.SEED0 + 100)
d = mp.Dmpc("bound", **kw)
p, v, a = d.init_batch(po, pf)
l = p.copy()
x_p = po.copy(); x_v = np.zeros_like(po); x_a = np.zeros_like(po)
tot = []
for k in range(1, 80):
    out = d.step_batch(l, x_p, x_v, x_a, pf)
    st, info = out["status"][0], out["info"][0]
    if (st != 1).any():
        print("step", k, "abort", np.unique(st)); break
    it = info[:, 4]; i = int(it.argmax())
    tot.append(int(it.max()))
    if it.max() > 12:
        heavy = np.where(it > 12)[0]
        print(f"step {k:3d}: max iters {it.max():4d} agent {i:3d} tries {info[i,2]} nv {info[i,1]} nslack {info[i,5]} nactive {info[i,6]} rows {info[i,7]} violk {info[i,0]} | heavy agents:",
              [(int(h), int(it[h]), int(info[h, 2]), int(info[h, 6])) for h in heavy][:6])
    l = out["p"].copy()
    x_p, x_v, x_a = out["p"][..., :3].copy(), out["v"][..., :3].copy(), out["a"][..., :3].copy()
tot = np.array(tot)
print("steps", len(tot), "sum of per-step max iterations", tot.sum(), "steps with max>12:", (tot > 12).sum(), "their share", tot[tot > 12].sum() / tot.sum())
