"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_trace_bench.py [n_agents]): the pivot sequence of infeasible agents of the headline
workload (first scene): which constraints does the dual method add while it proves infeasibility?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
cfg, N, S = wl.CONFIGS["C2"], 100, 4
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("hard", **kw)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
l, _, _ = d.init_batch(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
out = d.step_batch(l, xp, xv, xa, pf)
st = out["status"].reshape(-1); inf = out["info"].reshape(-1, 8)
cand = [g for g in np.where(st & 8)[0] if 15 <= inf[g, 4] <= 40][: int(sys.argv[1]) if len(sys.argv) > 1 else 3]
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["BH", "BL", "PH", "PL", "CO", "SU", "SL"]
cap = 64
for g in cand:
    sc, n = divmod(g, N)
    L.dmpc_debug_trace(d._ctx, n, cap, None)
    o = d.step_batch(l[sc], xp[sc], xv[sc], xa[sc], pf[sc])
    buf = np.zeros((cap, 8))
    L.dmpc_debug_trace(d._ctx, n, cap, buf.ctypes.data_as(C.c_void_p))
    r = d.rows_one(l[sc], n, xp[sc, n], xv[sc, n])
    print(f"agent {g}: status {o['status'][n]} iters {o['info'][n][4]} rows {o['info'][n][1]}; rows per step {np.bincount(r['kc'] - 1, minlength=15).tolist()}")
    seq = []
    for i, row in enumerate(buf[:cap - 4]):
        if row[3] == 0: break
        code = int(row[0]); ty = code >> 16; idx = code & 0xffff
        if ty < 4: seq.append(f"{names[ty]}k{idx // 3}a{idx % 3}")
        else: seq.append(f"CO{idx}(k{r['kc'][idx] - 1})" if idx < len(r["kc"]) else f"CO{idx}")
    print("   " + " ".join(seq))
