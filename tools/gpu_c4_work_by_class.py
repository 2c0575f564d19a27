"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_c4_work_by_class.py [mpc_step]): where the WORK of a C4 solve launch sits -- every agent's
solve duration (persistent waves, dmpc_debug_trace mode -5) grouped by what kind of agent it is: finished by the scan, no collision rows, rows on the first
ladder level, ladder climbers."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
N = 10000
kstep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc("bound", **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(kstep - 2):
    o = d.step_batch(l, xp, xv, xa, pf); ok = o["status"] == 1
    l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
cap = N * 2 // 8 + 8
assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
d.profile(True)
out = d.step_batch(l, xp, xv, xa, pf)
solve_ms, scan_ms, _ = d.profile_read2()
buf = np.zeros(cap * 8)
assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:N * 2].reshape(N, 2)
dur = t[:, 1] * 1e-2
inf, st = out["info"][0], out["status"][0]
rows, tries, it, maxq = inf[:, 1], inf[:, 2], inf[:, 4], inf[:, 7]
print(f"MPC step {kstep}: solve launch {solve_ms * 1e3:.0f} us, sum of the agents' durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / 1792:.0f} us per wave slot of 1792")
cls = [("finished by the scan (no solve)", dur == 0), ("no rows", (dur > 0) & (rows == 0)), ("rows, first ladder level", (dur > 0) & (rows > 0) & (tries <= 1)),
       ("ladder climbers", (dur > 0) & (tries > 1)), ("status != solved", st != 1)]
for name, m in cls:
    if m.any():
        print(f"  {name:34s} {m.sum():5d} agents  {dur[m].sum() / dur.sum() * 100:5.1f} % of the work  mean {dur[m].mean():6.1f} us  p90 {np.percentile(dur[m], 90):6.1f}  max {dur[m].max():6.1f}  iterations mean {it[m].mean():5.1f}  working set max mean {maxq[m].mean():4.1f}")
for lo, hi in ((0, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, 2000)):
    m = (dur >= lo) & (dur < hi) & (dur > 0)
    print(f"  {lo:4d}-{hi:4d} us: {m.sum():5d} agents, {dur[m].sum() / dur.sum() * 100:5.1f} % of the work")
