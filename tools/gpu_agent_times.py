"""development (library built with `make DEV_TRACE=1`): duration of every agent's solve in the persistent launch against its
iteration count and working-set size"""
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
cap = S * N * 2 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -3, cap, None) == 0
    out = d.step_batch(l, po, z, z, pf)
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -3, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:S * N * 2].reshape(S * N, 2)          # indexed by QUEUE POSITION
start = (t[:, 0] - t[:, 0].min()) * 1e-2        # us (100 MHz)
dur = t[:, 1] * 1e-2
print(f"per-position durations: mean {dur.mean():.1f} us, max {dur.max():.1f} us; last end {np.max(start + dur):.1f} us")
top = np.argsort(dur)[-8:]
print("longest solves: queue position", top, "start", np.round(start[top], 1), "dur", np.round(dur[top], 1))
late = np.argsort(start + dur)[-8:]
print("last to end: position", late, "start", np.round(start[late], 1), "dur", np.round(dur[late], 1))
it = out["info"][..., 4].ravel(); mq = out["info"][..., 7].ravel()
print("iterations: max", it.max(), " agents > 60 iterations:", (it > 60).sum(), " total", it.sum())
for lo, hi in ((0, 1), (1, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 200)):
    pass
# cost model: duration against iterations needs the agent of each position: the order is internal; use the sorted tails instead
print("sorted durations (us) p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f" % tuple(np.percentile(dur, [50, 90, 99, 99.9])))
print("sorted iterations     p50 %d p90 %d p99 %d p99.9 %d" % tuple(np.percentile(it, [50, 90, 99, 99.9])))
top = np.argsort(dur)[::-1][:40]
print("queue positions of the 40 longest solves:", np.sort(top))
print("their durations by position order:", np.round(dur[np.sort(top)]).astype(int))
for lo, hi in ((0, 2048), (2048, 4096), (4096, 8192), (8192, 16384), (16384, 51200)):
    d = dur[lo:hi]
    print(f"positions {lo}-{hi}: mean {d.mean():.1f} us, p99 {np.percentile(d, 99):.0f}, max {d.max():.0f}")
