"""development: 512 whole transitions (100 agents, solveSoftDMPCbound) for several numbers of batch parts (development option split_parts): how many
concurrent MPC loops keep the GPU busiest?"""
import os, sys, time, subprocess
if len(sys.argv) == 1:
    for parts in (1, 2, 4, 8, 16):
        subprocess.call([sys.executable, __file__, str(parts)], env=dict(os.environ, DMPC_DEBUG_OPTIONS=f"split_parts={parts}"))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = dict(wl.CONFIGS["C4"]); kw = wl.solver_kwargs(cfg, 100)
d = mp.Dmpc("bound", **kw)
po, pf = wl.make_scenes(cfg, 512, 100, wl.SEED0 + 100)
d.transition(po[:64], pf[:64], 10, cfg["error_tol"], histories=False)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); r = d.transition(po, pf, cfg["K_T"], cfg["error_tol"], histories=False); best = min(best, time.perf_counter() - t)
print(f"parts {sys.argv[1]}: 512 transitions {best*1e3:.1f} ms, {((r['K_T_used']-1)*100).sum()/best/1e6:.1f} M solves/s, completed {int(((r['scene_status'] & 256) != 0).sum())}")
