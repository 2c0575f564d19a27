"""development: iterations of the dual method against the size of the working set it ends with (the floor of a method that adds one constraint per
iteration) -- headline launch (hard) and the bound replay"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
for variant, kcap in (("hard", 1), ("bound", 12), ("ondemand", 12)):
    cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc(variant, **kw)
    l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, kcap, wl.SEED0 + 2)
    out = d.step_batch(l, xp, xv, xa, pf)
    st = out["status"].reshape(-1); inf = out["info"].reshape(-1, 8)
    ok = (st & 1) == 1; it = inf[:, 4]; q = inf[:, 6]; mq = inf[:, 7]
    live = ok & (it > 0)
    print(f"{variant}: solved {ok.sum()} (with iterations {live.sum()}): iterations mean {it[live].mean():.2f}, final working set mean {q[live].mean():.2f}, largest on the way {mq[live].mean():.2f}; "
          f"iterations - final set: mean {(it[live]-q[live]).mean():.2f} ({(it[live]-q[live]).sum()} of {it[live].sum()} iterations); not solved: {(~ok).sum()} agents, {it[~ok].sum()} iterations")
