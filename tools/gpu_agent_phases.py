"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_agent_phases.py variant scene agent ...): cycles by solver phase of chosen
agents of the bench's replay workload, each solved alone (one wave on the whole GPU): where does a long agent spend its time?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
variant = sys.argv[1]
gids = [x for x in sys.argv[2:]]   # agent ids, or `max`: the agent with the most iterations of the replayed step
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["pivot scan", "descriptor", "T T's", "resid+dir", "append/other", "drops", "verify", "ladder", "setup", "-", "-", "-", "-", "s", "T's", "ratio", "step+certs"]
cap = 8   # (the phase block takes rows cap-2 .. : 20 doubles)
if "max" in gids:
    o_all = d.step_batch(l, xp, xv, xa, pf)
    it_all = o_all["info"][..., 4].reshape(-1)
    gids = [int(i) for i in np.argsort(it_all)[-gids.count("max"):][::-1]] + [int(x) for x in gids if x != "max"]
gids = [int(x) for x in gids]
for g in gids:
    sc, n = divmod(g, N)
    for rep in range(2):
        L.dmpc_debug_trace(d._ctx, n, cap, None)
        out = d.step_batch(l[sc], xp[sc], xv[sc], xa[sc], pf[sc])
        buf = np.zeros((cap, 8))
        L.dmpc_debug_trace(d._ctx, n, cap, buf.ctypes.data_as(C.c_void_p))
    ph = buf.ravel()[(cap - 3) * 8:(cap - 3) * 8 + 20]
    i = out["info"][n]
    use = [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16]
    tot = ph[use].sum()
    print(f"agent {g} status {out['status'][n]} rows {i[1]} tries {i[2]} iters {i[4]} maxq {i[7]}: total {tot/100:.0f} us (100 MHz ticks)" if False else
          f"agent {g} status {out['status'][n]} rows {i[1]} tries {i[2]} iters {i[4]} maxq {i[7]}: total {tot:.0f} ticks; verifications {ph[10]:.0f} drops {ph[11]:.0f} certs {ph[12]:.0f}")
    print("   " + "  ".join(f"{names[u]} {ph[u]/tot*100:.0f}%" for u in use))
    print("   per iteration (ticks): " + "  ".join(f"{names[u]} {ph[u]/max(i[4],1):.0f}" for u in use))
