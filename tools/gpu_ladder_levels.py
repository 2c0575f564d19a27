"""development: iterations per ladder level of the multi-try agents of the bench's `solveSoftDMPCbound` replay launch (the same launch with
max_tries = 1, 2, 3, ...: the differences of the iteration counts are the levels) -- what would solving the levels of an agent side by
side take off the longest chain of the launch?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "bound"
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
full = d.step_batch(l, xp, xv, xa, pf)
inf = full["info"].reshape(-1, 8)
tries, iters = inf[:, 2], inf[:, 4]
per = []
for mt in range(1, 7):
    dm = mp.Dmpc(variant, **dict(kw, max_tries=mt))
    o = dm.step_batch(l, xp, xv, xa, pf)
    per.append(o["info"].reshape(-1, 8)[:, 4].copy())
per = np.array(per)          # [mt][agent] cumulative iterations with at most mt tries
lev = np.diff(np.vstack([np.zeros_like(per[0]), per]), axis=0)   # iterations of level mt
heavy = np.argsort(iters)[::-1][:40]
print("agent tries iters | per level (cumulative caps 1..6) | serial chain, chain with level L+1 started at iteration 20 of level L")
def chain(levels, spec):
    # levels: iterations of the levels actually run (zeros = certified/skipped); start of level i+1 = start_i + min(len_i, spec) if len_i >= spec else end_i
    t0 = 0; end = 0
    for n in levels:
        if n == 0: continue
        end = t0 + n
        t0 = t0 + spec if n >= spec else end
    return end
tot_serial = []; tot_spec = []
for a in heavy:
    lv = [int(x) for x in lev[:, a]]
    c = chain(lv, 20)
    tot_serial.append(iters[a]); tot_spec.append(c)
    print(f"{a:6d} {tries[a]:3d} {iters[a]:5d} | {lv} | {sum(lv)} -> {c}")
print("longest chain serial", max(tot_serial), "speculative", max(tot_spec))
for spec in (8, 12, 16, 20, 24, 32):
    cs = [chain([int(x) for x in lev[:, a]], spec) for a in heavy]
    extra = sum(1 for a in range(len(iters)) if iters[a] >= spec)
    print(f"spec after {spec:2d}: longest chain {max(cs)}; agents reaching it {extra}")
