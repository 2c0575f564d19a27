"""development (round 4): solve / scan kernel times of the replay launches in the four precisions (f64, mixed, f32factor, low)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
for variant, k_cap in (("hard", 12), ("bound", 12), ("repair", 12)):
    cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
    kw = wl.solver_kwargs(cfg, N)
    d0 = mp.Dmpc(variant, **kw)
    l, xp, xv, xa, pf, alive = bench.capture_state(d0, cfg, S, N, k_cap, wl.SEED0 + 2)
    for precision in ("f64", "mixed", "f32factor", "low"):
        d = mp.Dmpc(variant, precision=precision, **kw)
        for _ in range(3): out = d.step_batch(l, xp, xv, xa, pf)
        d.profile(True)
        for _ in range(10): out = d.step_batch(l, xp, xv, xa, pf)
        sms, cms, _ = d.profile_read2()
        d.profile(False)
        print(f"{variant:7s} {precision:10s}: solve {sms*1e3:7.1f} us  scan+order {cms*1e3:6.1f} us  -> {S*N/(sms+cms)/1e3:6.2f} M solves/s | iterations mean {out['info'][...,4].mean():.2f} solved {(out['status']&1).mean():.4f}")
