"""development (library built with `make DEV_TRACE=1`): when do the persistent waves of the solve launch start and end?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
cfg, N, S = wl.CONFIGS["C2"], 100, int(sys.argv[1]) if len(sys.argv) > 1 else 512
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
d = mp.Dmpc("hard", **kw)
l, _, _ = d.init_batch(po, pf)
z = np.zeros_like(po)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
NW = 3072            # persistent waves of the launch: 256 CUs x 12 (slack-free kernels since round 4; 9 in round 3; 8 for the slack variants)
cap = NW * 3 // 8 + 8
for rep in range(3):
    assert L.dmpc_debug_trace(d._ctx, -2, cap, None) == 0
    out = d.step_batch(l, po, z, z, pf)
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -2, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:NW * 3].reshape(NW, 3)
t0, t1, n = t[:, 0], t[:, 1], t[:, 2]
ok = n > 0
tick = 1e-8   # wall_clock64: 100 MHz
b, e = (t0[ok] - t0[ok].min()) * tick * 1e6, (t1[ok] - t0[ok].min()) * tick * 1e6
print(f"waves {ok.sum()}  start spread {b.max():.1f} us  end: min {e.min():.1f} median {np.median(e):.1f} p90 {np.percentile(e,90):.1f} max {e.max():.1f} us")
print(f"agents per wave: min {n[ok].min():.0f} median {np.median(n[ok]):.0f} max {n[ok].max():.0f}; busy fraction {((e-b).sum()/(ok.sum()*e.max())):.3f}")
last = np.argsort(e)[-5:]
print("last waves end at", np.round(e[last], 1), "agents", n[ok][last])
h, edges = np.histogram(e, bins=12)
print("wave ends histogram (us):", [f"{edges[i]:.0f}-{edges[i+1]:.0f}: {h[i]}" for i in range(len(h))])
