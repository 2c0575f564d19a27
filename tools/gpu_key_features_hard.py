"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_key_features_hard.py): the headline launch (C2, solveHardDMPC, 512 scenes): per agent
the scan's launch-order key next to the work the solve then did.  Saved to gpurun_out/key_features_hard.npz"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
cfg, N, S = wl.CONFIGS["C2"], 100, 512
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
d = mp.Dmpc("hard", **kw)
l, _, _ = d.init_batch(po, pf)
z = np.zeros_like(po)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
assert L.dmpc_debug_trace(d._ctx, -6, 8, None) == 0
out = d.step_batch(l, po, z, z, pf)
inf = out["info"].reshape(-1, 8); st = out["status"].reshape(-1)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/key_features_hard.npz", info=inf, status=st)
key = inf[:, 5] & 255; cost = inf[:, 3]; it = inf[:, 4]
print("agents", len(key), "infeasible", (st & 8).astype(bool).mean(), "cost mean", cost.mean(), "iters mean", it.mean())
for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 32), (32, 48), (48, 256)):
    m = (key >= lo) & (key < hi)
    if m.any(): print(f"key {lo:3d}-{hi:3d}: {m.sum():6d} agents, iterations mean {it[m].mean():6.1f} p99 {np.percentile(it[m], 99):6.1f} max {it[m].max():4d}; >= 50 iterations: {(it[m] >= 50).sum()}")
print("correlation(iterations, key)", np.corrcoef(it, key)[0, 1])
# (round 4: the acceleration bounds violated at the unconstrained minimiser were exported here too -- 10 253 of the 11 079 agents with key < 24 have none and
# still take up to 28 iterations: no help for the low end of the queue, where the launch's last 100 us come from)
