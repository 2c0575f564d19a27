"""development aid: find and trace the slowest agent of the bench's secondary (bound) workload."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "bound"
kcap = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = dict(wl.CONFIGS["C2"], variant=variant)
S, N = 64, 100
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, kcap, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
it = out["info"][..., 4]
s, n = np.unravel_index(it.argmax(), it.shape)
print("slowest agent: scene", s, "agent", n, "info", out["info"][s, n], "status", out["status"][s, n])
print("iters histogram:", np.histogram(it, bins=[0, 1, 2, 5, 10, 20, 50, 100, 200, 1000])[0])
cap = 420
L = d._L
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.dmpc_debug_trace(d._ctx, int(n), cap, None)
o1 = d.step_batch(l[s], xp[s], xv[s], xa[s], pf[s])
buf = np.zeros((cap, 8))
L.dmpc_debug_trace(d._ctx, int(n), cap, buf.ctypes.data_as(C.c_void_p))
names = ["BH", "BL", "PH", "PL", "CO", "SU", "SL"]
from collections import Counter
types = Counter(); prev = None; partial = 0
for i, r in enumerate(buf[:-2]):
    if r[3] == 0: break
    code = int(r[0]); types[names[code >> 16]] += 1
    if prev == code: partial += 1
    prev = code
    if i < 40 or i % 25 == 0:
        print(f"{i+1:4d} p={names[code>>16]}{code&0xffff:<4d} q={int(r[1]):2d} delta/spp={r[2]/r[3]:.2e} t1={r[4]:.3e} t2={r[5]:.3e} vp={r[6]:.2e}")
print("constraint types picked:", dict(types), "repeat (partial-step) iterations:", partial)
