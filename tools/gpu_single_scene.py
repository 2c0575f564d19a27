"""development: wall time per MPC step of ONE 100-agent scene (the literal configs[1] / configs[3] shapes), closed loop on the device"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
for cfgname, variant in (("C4", "bound"), ("C2", "hard")):
    cfg = wl.CONFIGS[cfgname]
    kw = wl.solver_kwargs(cfg, 100)
    po, pf = wl.make_scenes(cfg, 1, 100, wl.SEED0 + 100)
    d = mp.Dmpc(variant, **kw)
    d.transition(po, pf, 10, cfg["error_tol"])
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); r = d.transition(po, pf, cfg["K_T"], cfg["error_tol"], histories=False); dt = time.perf_counter() - t
        best = min(best, dt)
    steps = int(r["K_T_used"][0]) - 1
    print(f"{cfgname} {variant}: {steps} MPC steps, {best * 1e3:.2f} ms, {best / max(steps, 1) * 1e6:.1f} us per step, status {int(r['scene_status'][0])}")
