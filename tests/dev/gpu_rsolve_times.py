#!/usr/bin/env python3
"""Development (RSOLVE_TRACE build): start and duration of every agent of one solve launch of the reduced solver (C4-like scene, MPC step `step`)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiagent_planning_amd as mp
from multiagent_planning_amd import _lib, workload as wl
from oracle import oracle as orc
N, step, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = dict(wl.CONFIGS["C4"]); cfg["N"] = N
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, seed); po, pf = po[0], pf[0]
d = mp.Dmpc("bound", device=0, **kw)
l, _, _ = d.init_batch(po[None], pf[None]); l = l[0]
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(2, step):
    out = d.step_batch(l, xp, xv, xa, pf)
    ok = (out["status"] & 1) == 1
    l = np.where(ok[:, None], out["p"], l); xp = np.where(ok[:, None], out["p"][:, :3], xp); xv = np.where(ok[:, None], out["v"][:, :3], xv); xa = np.where(ok[:, None], out["a"][:, :3], xa)
L = _lib.load(); L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
cap = (2 * N + 7) // 8 + 1
assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
out = d.step_batch(l, xp, xv, xa, pf)
buf = np.zeros((cap, 8)); assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.ravel()[:2 * N].reshape(N, 2)
ran = t[:, 1] > 0
t0 = t[ran, 0].min()
start, dur = (t[ran, 0] - t0) / 100.0, t[ran, 1] / 100.0   # microseconds (100 MHz)
eq = out["info"][ran, 4]
print(f"agents solved in the kernel {ran.sum()} of {N}; launch span {(start + dur).max():.0f} us; sum of durations {dur.sum() / 1e3:.1f} ms; mean {dur.mean():.1f} us, p50 {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}")
print(f"us per EQP: total {dur.sum() / eq.sum():.2f}; agents with 1 EQP: {np.mean(dur[eq <= 1]) if (eq<=1).any() else 0:.1f} us; fit dur = a + b eqps:", np.polyfit(eq, dur, 1))
order = np.argsort(-(start + dur))[:10]
print("last to end (end, start, dur, eqps):", [(round(float(start[i] + dur[i])), round(float(start[i])), round(float(dur[i])), int(eq[i])) for i in order])
print("start times: p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.median(start), np.percentile(start, 90), np.percentile(start, 99), start.max()))
