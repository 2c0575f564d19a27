"""development aid: a batch of whole trials (transition + post-checks) of the reference's primary variant."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import driver, workload as wl
cfg = dict(wl.CONFIGS["C4"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for S in [int(x) for x in (sys.argv[2:] or ["1", "8", "64", "512"])]:
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc("bound", **kw)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 100)
    driver.run_trial(d, po[:1], pf[:1], 12, cfg["error_tol"])
    t0 = time.perf_counter()
    tr = d.transition(po, pf, cfg["K_T"], cfg["error_tol"])
    t1 = time.perf_counter()
    res = driver.run_trial(d, po, pf, cfg["K_T"], cfg["error_tol"])
    t2 = time.perf_counter()
    used = tr["K_T_used"]
    print(f"N={N} S={S:4d}: transition {1e3*(t1-t0):8.1f} ms ({1e3*(t1-t0)/S:7.3f} ms each); trial (transition+post-checks) {1e3*(t2-t1):8.1f} ms; "
          f"success {res['success'].mean():.3f} feasible {res['feasible'].mean():.3f} violation {res['violation'].mean():.3f}; "
          f"{((used-1)*N).sum()/(t1-t0)/1e6:.2f} M solves/s; mean steps {used.mean():.1f}")
