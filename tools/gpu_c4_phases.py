"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_c4_phases.py [agent ...]): cycles by solver phase of chosen agents of the
10^4-agent scene (C4, third MPC step), each traced inside the full launch: where does the solve launch of a large scene spend its time?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
N = 10000
gids = [int(x) for x in sys.argv[1:] if x != "top"] or [17, 1234, 5000, 7777, 9001]
pick_top = "top" in sys.argv[1:]
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc(cfg["variant"], **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(2):
    out = d.step_batch(l, xp, xv, xa, pf)
    ok = out["status"] == 1
    l = np.where(ok[..., None], out["p"], l); xp = np.where(ok[..., None], out["p"][..., :3], xp)
    xv = np.where(ok[..., None], out["v"][..., :3], xv); xa = np.where(ok[..., None], out["a"][..., :3], xa)
if pick_top:   # (round 4) the agents with the most iterations of this step
    o_ = d.step_batch(l, xp, xv, xa, pf)
    gids = [int(g) for g in np.argsort(o_["info"][0, :, 4])[-5:][::-1]]
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["pivot scan", "descriptor", "T T's / crash round", "resid+dir", "append/other", "drops", "verify", "ladder", "setup", "-", "-", "-", "-", "s", "T's", "ratio", "step+certs", "crash batch"]
cap = 8
tots = np.zeros(20); nag = 0
for g in gids:
    for rep in range(2):
        L.dmpc_debug_trace(d._ctx, g, cap, None)
        out = d.step_batch(l, xp, xv, xa, pf)
        buf = np.zeros((cap, 8))
        L.dmpc_debug_trace(d._ctx, g, cap, buf.ctypes.data_as(C.c_void_p))
    ph = buf.ravel()[(cap - 3) * 8:(cap - 3) * 8 + 20]
    i = out["info"][0, g]
    use = [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17]
    tot = ph[use].sum()
    if tot <= 0: print("agent", g, "not traced (status", out["status"][0, g], ")"); continue
    tots += ph; nag += 1
    print(f"agent {g} status {out['status'][0, g]} rows {i[1]} tries {i[2]} iters {i[4]} maxq {i[7]}: total {tot/100:.0f} us; verifications {ph[10]:.0f} drops {ph[11]:.0f} certs {ph[12]:.0f}")
    print("   " + "  ".join(f"{names[u]} {ph[u]/tot*100:.0f}%" for u in use if ph[u] > 0))
if nag:
    use = [0, 1, 2, 3, 4, 5, 6, 7, 8, 13, 14, 15, 16, 17]
    tot = tots[use].sum()
    print(f"mean of {nag} agents: {tot/nag/100:.0f} us; " + "  ".join(f"{names[u]} {tots[u]/tot*100:.0f}%" for u in use if tots[u] > 0))
