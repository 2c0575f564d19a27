"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_agent_phases.py variant scene agent ...): cycles by solver phase of chosen
agents of the bench's replay workload, each solved alone (one wave on the whole GPU): where does a long agent spend its time?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
variant = sys.argv[1]
gids = [int(x) for x in sys.argv[2:]]
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["pivot scan", "descriptor", "matvecs", "resid+dir", "ratio/step/append", "drops", "verify", "ladder", "setup"]
cap = 8
for g in gids:
    sc, n = divmod(g, N)
    for rep in range(2):
        L.dmpc_debug_trace(d._ctx, n, cap, None)
        out = d.step_batch(l[sc], xp[sc], xv[sc], xa[sc], pf[sc])
        buf = np.zeros((cap, 8))
        L.dmpc_debug_trace(d._ctx, n, cap, buf.ctypes.data_as(C.c_void_p))
    ph = buf.ravel()[(cap - 2) * 8:(cap - 2) * 8 + 13]
    i = out["info"][n]
    tot = ph[:9].sum()
    print(f"agent {g} status {out['status'][n]} rows {i[1]} tries {i[2]} iters {i[4]} maxq {i[7]}: total {tot/100:.0f} us (100 MHz ticks)" if False else
          f"agent {g} status {out['status'][n]} rows {i[1]} tries {i[2]} iters {i[4]} maxq {i[7]}: total {tot:.0f} ticks; verifications {ph[10]:.0f} drops {ph[11]:.0f} certs {ph[12]:.0f}")
    print("   " + "  ".join(f"{nm} {v/tot*100:.0f}%" for nm, v in zip(names, ph[:9])))
    print("   per iteration (ticks): " + "  ".join(f"{nm} {v/max(i[4],1):.0f}" for nm, v in zip(names, ph[:9])))
