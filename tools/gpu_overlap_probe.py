"""development aid: would overlapping sub-batches of scenes on separate streams speed up batched transitions?
G contexts (one HIP stream each) run S/G scenes each from G host threads concurrently."""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = dict(wl.CONFIGS["C4"]); N = 100; S = 512
kw = wl.solver_kwargs(cfg, N)
d0 = mp.Dmpc("bound", **kw)
po, pf = wl.make_scenes_device(d0, cfg, S, N, wl.SEED0 + 100)
for G in (1, 2, 4, 8):
    ds = [mp.Dmpc("bound", **kw) for _ in range(G)]
    sl = [slice(g * S // G, (g + 1) * S // G) for g in range(G)]
    for g in range(G): ds[g].transition(po[sl[g]][:2], pf[sl[g]][:2], 12, cfg["error_tol"], histories=False)
    def work(g): ds[g].transition(po[sl[g]], pf[sl[g]], 151, cfg["error_tol"], histories=False)
    for rep in range(2):
        ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
    print(f"{G} concurrent sub-batches of {S//G} scenes: {dt*1e3:.1f} ms for {S} transitions")
