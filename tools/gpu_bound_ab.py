"""development: solve / scan kernel times of the bench's `solveSoftDMPCbound` replay launch (python tools/with_lib.py <lib> tools/gpu_bound_ab.py for another build)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "bound"
precision = sys.argv[2] if len(sys.argv) > 2 else "f64"
k_cap = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, (int(sys.argv[4]) if len(sys.argv) > 4 else 512)
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, precision=precision, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, k_cap, wl.SEED0 + 2)
for rep in range(3):
    for _ in range(3): out = d.step_batch(l, xp, xv, xa, pf)
    d.profile(True)
    for _ in range(10): out = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    d.profile(False)
    inf = out["info"].reshape(-1, 8)
    print(f"{variant} {precision} step {k_cap}: solve {sms*1e3:7.1f} us  scan+order {cms*1e3:6.1f} us  | iterations mean {inf[:,4].mean():.2f} max {inf[:,4].max()}  tries max {inf[:,2].max()}")
