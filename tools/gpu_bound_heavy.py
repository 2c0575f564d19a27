"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_bound_heavy.py [variant]): the longest agents of the bench's
replay launch: duration against iterations, tries, rows, working set -- what does the tail of the launch consist of?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "bound"
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
cap = S * N * 2 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
    d.profile(True)
    out = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:S * N * 2].reshape(S * N, 2)
start = (t[:, 0] - t[:, 0].min()) * 1e-2; dur = t[:, 1] * 1e-2
inf = out["info"].reshape(-1, 8); st = out["status"].ravel()
print(f"{variant}: solve {sms*1e3:.0f} us scan {cms*1e3:.0f} us; sum(dur)/2048 = {dur.sum()/2048:.0f} us; last end {np.max(start+dur):.0f} us")
print("agent  start   dur  status viol_k rows tries case iters nslack nact maxq  us/iter")
for a in np.argsort(dur)[::-1][:25]:
    i = inf[a]
    print(f"{a:6d} {start[a]:6.0f} {dur[a]:6.0f} {st[a]:4d} {i[0]:5d} {i[1]:5d} {i[2]:4d} {i[3]:4d} {i[4]:6d} {i[5]:5d} {i[6]:4d} {i[7]:4d}  {dur[a]/max(i[4],1):.2f}")
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
    m = (inf[:, 4] >= lo) & (inf[:, 4] < hi)
    if m.any():
        print(f"iters {lo:3d}-{hi:4d}: {m.sum():6d} agents, mean dur {dur[m].mean():7.1f} us, total {dur[m].sum()/2048:6.1f} us/wave-slot, mean rows {inf[m,1].mean():.1f}")
