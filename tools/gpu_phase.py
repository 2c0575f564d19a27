"""development aid (phase cycle counts need a library built with `make -C multiagent_planning_amd/csrc DEV_TIMERS=1`): phase cycle counts of typical agents of the C2 bench workload (fixed overhead vs iterations)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "hard"
cfg = dict(wl.CONFIGS["C2"], variant=variant)
S, N = 8, 100
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
it = out["info"][0, :, 4]
order = np.argsort(it)
L = d._L
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
cap = 16
for n in [order[5], order[N // 4], order[N // 2], order[3 * N // 4], order[-5], order[-1]]:
    L.dmpc_debug_trace(d._ctx, int(n), cap, None)
    d.step_batch(l[0], xp[0], xv[0], xa[0], pf[0])
    buf = np.zeros((cap, 8))
    L.dmpc_debug_trace(d._ctx, int(n), cap, buf.ctypes.data_as(C.c_void_p))
    ph = buf[cap - 1]; p2 = buf[cap - 2]
    print(f"agent {n:3d} iters {int(ph[7]):3d} rows {out['info'][0, n, 1]:3d} status {out['status'][0, n]} | hdr {int(ph[0]):6d} setup {int(ph[1]):6d} solve {int(ph[2]):7d} out {int(ph[3]):5d} | "
          f"viol {int(ph[4]):6d} matvec {int(ph[5]):6d} nu {int(ph[6]):6d} ratio {int(p2[0]):6d} step {int(p2[1]):6d} desc {int(p2[2]):6d} drop {int(p2[3]):6d}"
          + (f" | per-iter {ph[2]/ph[7]:.0f}" if ph[7] else ""))
