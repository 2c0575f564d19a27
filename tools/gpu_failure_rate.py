"""development aid: the reference's failure-rate experiment (test/failure_rate.m: success probability of
solveSoftDMPCbound transitions vs swarm size at constant density) with S random trials per size, next to the success
rates recorded in data/failure_rate/failure_rate2.mat (50 MATLAB trials per size)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import driver, workload as wl
RECORDED = {20: 1.0, 40: 1.0, 60: 1.0, 80: 0.96, 100: 0.94, 120: 0.74, 140: 0.66, 160: 0.62, 180: 0.40, 200: 0.28}
cfg = dict(wl.CONFIGS["C4"])
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for N in sorted(RECORDED):
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc("bound", **kw)
    t0 = time.perf_counter()
    po, pf = wl.make_scenes_device(d, cfg, S, N, wl.SEED0 + 7 * N) if os.environ.get("DEVICE_SCENES", "1") == "1" else wl.make_scenes(cfg, S, N, wl.SEED0 + 7 * N)
    t1 = time.perf_counter()
    res = driver.run_trial(d, po, pf, cfg["K_T"], cfg["error_tol"], histories=False)
    t2 = time.perf_counter()
    p = res["success"].mean()
    se = np.sqrt(max(p * (1 - p), 1e-9) / S)
    print(f"N={N:3d}: success {p:.3f} +- {se:.3f} (recorded {RECORDED[N]:.2f} over 50 trials)  feasible {res['feasible'].mean():.3f} "
          f"failed_goal {res['failed_goal'].mean():.3f} violation {res['violation'].mean():.3f}  [{S} trials: {1e3*(t2-t1):.0f} ms GPU, {1e3*(t1-t0):.0f} ms scene sampling]", flush=True)
