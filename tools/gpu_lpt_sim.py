"""development aid: how much of the solve launch is scheduling?  list-scheduling simulation of the headline workload on
2048 wave slots with different launch orders (cost model: 1.5 + iterations, in units of one iteration)."""
import sys, os, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
cfg = wl.CONFIGS["C2"]; N = 100; S = 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("hard", **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
it = out["info"][..., 4].reshape(-1).astype(float); nr = out["info"][..., 1].reshape(-1)
cost = 1.5 + it
def sched(order, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in cost[order]:
        t = heapq.heappop(h); heapq.heappush(h, t + c)
    return max(h)
n = len(cost)
print("agents", n, "sum/slots", cost.sum() / 2048, "max single", cost.max(), "corr(rows, iters)", np.corrcoef(nr, it)[0, 1])
rng = np.random.default_rng(0)
print("random order       :", sched(rng.permutation(n)))
print("by rows (current)  :", sched(np.argsort(-nr, kind="stable")))
print("by iterations (LPT):", sched(np.argsort(-it, kind="stable")))
