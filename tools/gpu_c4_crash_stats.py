"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_c4_crash_stats.py [mpc_step]): the crash start of the 10^4-agent scene in numbers --
per agent the appends without a step by source (factor table / three-axes products / one at a time inside the iteration loop), crash rounds, passes that
dropped negative multipliers, and the real iterations that follow (dmpc_debug_trace mode -4)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
N = 10000
kstep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc("bound", **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(kstep - 2):
    o = d.step_batch(l, xp, xv, xa, pf); ok = o["status"] == 1
    l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
assert L.dmpc_debug_trace(d._ctx, -4, 8, None) == 0
inf = d.step_batch(l, xp, xv, xa, pf)["info"][0]
nfast, rounds, negp, tbl, gen, it = inf[:, 0], inf[:, 1], inf[:, 3], inf[:, 5], inf[:, 6], inf[:, 4]
solved = it > 0
inloop = nfast - tbl - gen
print(f"MPC step {kstep}: {solved.sum()} agents through the solver; iterations mean {it[solved].mean():.1f} = appended without a step {nfast[solved].mean():.1f} "
      f"(table {tbl[solved].mean():.1f}, three-axes products {gen[solved].mean():.1f}, one at a time in the loop {inloop[solved].mean():.1f}) + real {(it - nfast)[solved].mean():.1f}")
print(f"   crash rounds mean {rounds[solved].mean():.2f}; agents with a negative-multiplier pass {(negp[solved] > 0).mean():.3f} (passes mean {negp[solved].mean():.2f})")
for lo, hi in ((0, 1), (1, 3), (3, 6), (6, 12), (12, 64)):
    m = solved & (inloop >= lo) & (inloop < hi)
    print(f"   in-loop appends {lo:2d}-{hi - 1:2d}: {m.sum():5d} agents, real iterations mean {(it - nfast)[m].mean() if m.any() else 0:.1f}")
