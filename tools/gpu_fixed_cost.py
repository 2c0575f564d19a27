"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_fixed_cost.py): per-agent time of the headline solve launch under iteration caps -- what do
set-up + one violation scan + outputs cost an agent (cap 0, no crash start), what the crash batch with its multiplier solve (cap 0), what the whole solve?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
N = 10000
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.dmpc_debug_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
cap = N * 2 // 8 + 8
for label, opts in (("set-up + scan + outputs", {"iter_cap": 0, "crash_min": 99}), ("+ crash batch and its multiplier solve", {"iter_cap": 0}), ("whole solve", {})):
    d = mp.Dmpc(cfg["variant"], **kw)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(2):   # the state of MPC step 4
        o = d.step_batch(l, xp, xv, xa, pf); ok = o["status"] == 1
        l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
        xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
    for k, v in opts.items(): assert L.dmpc_debug_option(d._ctx, k.encode(), v) == 0
    for rep in range(3):
        assert L.dmpc_debug_trace(d._ctx, -3, cap, None) == 0
        d.profile(True)
        out = d.step_batch(l, xp, xv, xa, pf)
        solve_ms, scan_ms, _ = d.profile_read2()
        buf = np.zeros(cap * 8)
        assert L.dmpc_debug_trace(d._ctx, -3, cap, buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf[:N * 2].reshape(N, 2)
    ran = t[:, 1] > 0
    start = (t[ran, 0] - t[ran, 0].min()) * 1e-2; dur = t[ran, 1] * 1e-2
    print(f"{label}: solve launch {solve_ms*1e3:.0f} us; {ran.sum()} positions; duration mean {dur.mean():.1f} p10 {np.percentile(dur,10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} us; "
          f"first start {start.min():.1f}, last end {(start+dur).max():.1f} us; sum / 1792 slots = {dur.sum()/1792:.0f} us")

# the waves of the whole-solve launch: when does each end, how many agents did it solve (dmpc_debug_trace agent -2)
capw = 2048 * 3 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -2, capw, None) == 0
    out = d.step_batch(l, xp, xv, xa, pf)
    bufw = np.zeros(capw * 8)
    assert L.dmpc_debug_trace(d._ctx, -2, capw, bufw.ctypes.data_as(C.c_void_p)) == 0
w = bufw[:2048 * 3].reshape(2048, 3)
w = w[w[:, 2] > 0]
t0 = w[:, 0].min()
beg = (w[:, 0] - t0) * 1e-2; end = (w[:, 1] - t0) * 1e-2
print(f"waves that solved agents: {len(w)}; begin p50 {np.median(beg):.1f} max {beg.max():.1f} us; end p10 {np.percentile(end,10):.0f} p50 {np.median(end):.0f} p90 {np.percentile(end,90):.0f} max {end.max():.0f} us; "
      f"agents per wave mean {w[:,2].mean():.2f} min {w[:,2].min():.0f} max {w[:,2].max():.0f}; busy share of the waves' own spans: {dur.sum() / (end - beg).sum():.3f}")

# the tail: which queue positions end last, and how well does the queue order (heaviest first by the scan's key) predict the durations?
assert L.dmpc_debug_trace(d._ctx, -3, cap, None) == 0
out = d.step_batch(l, xp, xv, xa, pf)
buf = np.zeros(cap * 8)
assert L.dmpc_debug_trace(d._ctx, -3, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:N * 2].reshape(N, 2); ran = np.nonzero(t[:, 1] > 0)[0]
st = (t[ran, 0] - t[ran, 0].min()) * 1e-2; du = t[ran, 1] * 1e-2; en = st + du
last = np.argsort(en)[-12:][::-1]
print("last to end (queue position, start, duration us):", [(int(ran[i]), int(st[i]), int(du[i])) for i in last])
for lo, hi in ((0, 256), (256, 1792), (1792, 3584), (3584, 6000), (6000, 8000), (8000, 10000)):
    m = (ran >= lo) & (ran < hi)
    if m.any(): print(f"   positions {lo:5d}-{hi:5d}: start mean {st[m].mean():6.0f} us, duration mean {du[m].mean():6.1f} p90 {np.percentile(du[m],90):6.1f} max {du[m].max():6.1f} us")

# list scheduling of the measured durations on 1792 slots: the queue's order, the perfect order (longest first), the reverse, and what the
# scan's key is worth against a perfect one (durations as measured in this launch; contention effects ignored)
import heapq
def makespan(seq, slots=1792):
    h = [0.0] * slots; heapq.heapify(h)
    for x in seq: heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)
order_q = np.argsort(ran)   # queue order = position order
print(f"list scheduling of the measured durations: queue order {makespan(du[order_q]):.0f} us, longest first {makespan(np.sort(du)[::-1]):.0f} us, random {makespan(np.random.default_rng(1).permutation(du)):.0f} us; "
      f"sum / slots {du.sum()/1792:.0f} us, longest agent {du.max():.0f} us; measured launch end {en.max():.0f} us")
it = out["info"][0, :, 4]
