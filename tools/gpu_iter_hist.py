"""development: distribution of the active-set iterations of the bench workload (C2 hard, 512 scenes) by outcome"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg, N, S = wl.CONFIGS["C2"], 100, 512
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
d = mp.Dmpc("hard", **kw)
l, _, _ = d.init_batch(po, pf)
z = np.zeros_like(po)
out = d.step_batch(l, po, z, z, pf)
st, it, mq = out["status"].ravel(), out["info"][..., 4].ravel(), out["info"][..., 7].ravel()
tot = it.sum()
for name, m in (("solved", (st & 1) == 1), ("infeasible", (st & 8) != 0)):
    x = it[m]
    print(f"{name}: {m.sum()} agents ({m.mean():.3f}), iterations: share {x.sum() / tot:.3f} mean {x.mean():.2f} median {np.median(x):.0f} p90 {np.percentile(x, 90):.0f} p99 {np.percentile(x, 99):.0f} max {x.max()}; zero-iteration agents {np.mean(x == 0):.3f}")
print("iteration histogram (all):", np.bincount(np.minimum(it, 60) // 4)[:16], "(bins of 4)")
print("peak working set > 32:", np.mean(mq > 32), " > 40:", np.mean(mq > 40), "max", mq.max())
print("rows built: mean", out["info"][..., 1].mean())
for thr in (24, 40, 60, 80):
    m = it > thr
    print(f"> {thr} iterations: {m.sum()} agents, infeasible {((st[m] & 8) != 0).sum()}, solved {((st[m] & 1) != 0).sum()}, share of all iterations {it[m].sum() / tot:.3f}")
