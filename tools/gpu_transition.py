"""development aid: whole transitions on the device (dmpc_transition) for the reference's recorded cases."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = dict(wl.CONFIGS["C4"])   # failure_rate.m constants (solveSoftDMPCbound), density-scaled box
for N, S in [(20, 8), (100, 8), (200, 8), (100, 1)]:
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc("bound", **kw)
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + N)
    d.transition(po[:1], pf[:1], 10, cfg["error_tol"])   # warm-up (allocations, module load)
    t0 = time.perf_counter()
    res = d.transition(po, pf, cfg["K_T"], cfg["error_tol"])
    dt = time.perf_counter() - t0
    used = res["K_T_used"]; ok = (res["scene_status"] & 1) == 1
    reached = [np.linalg.norm(res["pk"][s][:, used[s] - 1] - pf[s], axis=1).max() < cfg["error_tol"] for s in range(S)]
    nsolve = int(((used - 1) * N).sum())
    print(f"N={N:4d} S={S}: wall {dt*1e3:8.1f} ms for {S} transitions ({dt/S*1e3:.1f} ms each), MPC steps {used.tolist()}, "
          f"completed(no abort)={int(ok.sum())}/{S} reached_goal={int(np.sum(reached))}/{S}, {nsolve/dt/1e6:.2f} M useful solves/s")
