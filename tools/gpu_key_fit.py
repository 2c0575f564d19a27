"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_key_fit.py): measured solve durations of the headline launch against what the scan knows
about an agent (steps with a violated row, tightness class, row count): which combination orders the queue best?
(round 3: none beats the key in use -- 920 us simulated against 858 for the measured durations; the count of violated rows and the largest /
summed relative violation at the unconstrained minimiser correlate with the duration at 0.22-0.29, the key at 0.39)"""
import sys, os, ctypes as C, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
cfg, N, S = wl.CONFIGS["C2"], 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("hard", **kw)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
l, _, _ = d.init_batch(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.dmpc_debug_read_hdr.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
T = S * N; cap = T * 2 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
    out = d.step_batch(l, xp, xv, xa, pf)
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
dur = buf[:T * 2].reshape(T, 2)[:, 1] * 1e-2
hdr = np.zeros((T, 8), np.int32)
assert L.dmpc_debug_read_hdr(d._ctx, hdr.ctypes.data_as(C.c_void_p), T) == 0
key = hdr[:, 7] & 255; nr = hdr[:, 0]; steps = key >> 2; tight = key & 3
st = out["status"].reshape(-1); it = out["info"].reshape(-1, 8)[:, 4]
print("corr with duration: key %.3f steps %.3f tight %.3f rows %.3f iterations %.3f" % tuple(np.corrcoef(x, dur)[0, 1] for x in (key, steps, tight, nr, it)))
X = np.stack([steps, tight, nr, steps * nr, np.ones(T)], 1).astype(float)
w, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("least squares duration ~ %.2f steps + %.2f tight + %.3f rows + %.4f steps*rows + %.1f" % tuple(w))
def makespan(order, slots=2304):
    hh = [0.0] * slots; heapq.heapify(hh)
    for c in dur[order]:
        t = heapq.heappop(hh); heapq.heappush(hh, t + c)
    return max(hh)
cands = {"key in use (4 steps + tight)": key, "+ rows/64": key + (nr >> 6), "+ rows/32": key + (nr >> 5), "+ rows/16": key + (nr >> 4), "rows": nr,
         "least squares": X @ w, "2 steps + tight + rows/32": 2 * steps + tight + (nr >> 5), "steps*rows": steps * nr,
         "measured duration (bound)": dur, "iterations": it}
print("list-scheduling makespan on 2304 slots (sum/slots = %.0f us):" % (dur.sum() / 2304))
for name, k in cands.items():
    print(f"  {name:34s} {makespan(np.argsort(-k, kind='stable')):7.1f} us")
for lo, hi in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64)):
    m = (key >= lo) & (key < hi)
    if m.any(): print(f"key {lo:2d}-{hi:2d}: {m.sum():6d} agents, duration mean {dur[m].mean():6.1f} p90 {np.percentile(dur[m],90):6.1f} p99 {np.percentile(dur[m],99):6.1f} us; rows mean {nr[m].mean():.0f}; infeasible {((st[m] & 8) != 0).mean():.3f}")
