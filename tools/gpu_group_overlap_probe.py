"""development: the single-process multi-GPU path with the ranks emulated on ONE GPU -- whole transitions of a batch as one group and as two
groups side by side (the exchange of one half under the solve of the other).  The copies are same-device copies here: what this shows is
the host-side cost of the exchange protocol (threads, barriers, events), not xGMI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = dict(wl.CONFIGS["C4"])
for G, N, S in ((1, 200, 128), (2, 200, 128), (4, 200, 128), (8, 200, 128)):
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 60)
    for parts in (1, 2):
        mp.Dmpc.emulate_devices(G if G > 1 else 0)
        d = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL if G > 1 else 0, **kw)
        d.debug_option("no_split", 1) if parts == 1 else None
        d.transition(po[:8], pf[:8], 6, cfg["error_tol"], histories=False)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); r = d.transition(po, pf, 40, cfg["error_tol"], histories=False); best = min(best, time.perf_counter() - t)
        steps = int((r["K_T_used"] - 1).max())
        print(f"G={G} ranks (emulated) parts={parts}: {S} scenes x {N} agents, {steps} MPC steps in {best*1e3:.1f} ms = {best/steps*1e3:.3f} ms per step")
        del d
mp.Dmpc.emulate_devices(0)
