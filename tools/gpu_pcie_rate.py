"""development aid: PCIe-inclusive rate of the host-buffer entry point (dmpc_step_batch) on the headline workload."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
cfg = wl.CONFIGS["C2"]; N = 100
for S in (64, 512):
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc("hard", **kw)
    l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
    for _ in range(3): d.step_batch(l, xp, xv, xa, pf)
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps): d.step_batch(l, xp, xv, xa, pf)
    dt = (time.perf_counter() - t0) / reps
    print(f"S={S}: dmpc_step_batch (host buffers in and out, pageable numpy memory) {dt*1e3:.2f} ms per step = {S*N/dt/1e6:.2f} M solves/s")
