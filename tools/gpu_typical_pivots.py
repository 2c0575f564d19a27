"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_typical_pivots.py [mpc_step] [n]): the pivot sequences of n TYPICAL agents of the
10^4-agent scene (evenly spaced quantiles of the iteration count among the agents the solver ran) -- what are the real iterations of a mean agent?"""
import sys, os, ctypes as C, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
kstep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nag = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = 10000
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc("bound", **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(kstep - 2):
    out = d.step_batch(l, xp, xv, xa, pf)
    ok = out["status"] == 1
    l = np.where(ok[..., None], out["p"], l); xp = np.where(ok[..., None], out["p"][..., :3], xp)
    xv = np.where(ok[..., None], out["v"][..., :3], xv); xa = np.where(ok[..., None], out["a"][..., :3], xa)
o_ = d.step_batch(l, xp, xv, xa, pf)
it = o_["info"][0, :, 4]
ran = np.nonzero(it > 0)[0]
srt = ran[np.argsort(it[ran])]
picks = [int(srt[int((j + 0.5) / nag * len(srt))]) for j in range(nag)]
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["BH", "BL", "PH", "PL", "CO", "SU", "SL"]
cap = 200
tot = collections.Counter()
for g in picks:
    L.dmpc_debug_trace(d._ctx, g, cap, None)
    out = d.step_batch(l, xp, xv, xa, pf)
    buf = np.zeros((cap, 8))
    L.dmpc_debug_trace(d._ctx, g, cap, buf.ctypes.data_as(C.c_void_p))
    i = out["info"][0, g]
    seq = []
    for r in buf[:cap - 12]:
        if r[3] == 0: continue
        code = int(r[0]); ty, idx = code >> 16, code & 0xffff
        nm = f"{names[ty]}{idx // 3}{'xyz'[idx % 3]}" if ty < 4 else f"{names[ty]}{idx}"
        full = r[5] <= r[4]
        seq.append(f"{nm}{'+' if full else '-'}@{int(r[1])}")
        tot[(names[ty], '+' if full else '-')] += 1
    print(f"agent {g}: viol_k {i[0]} rows {i[1]} tries {i[2]} iters {i[4]} nslack {i[5]} active {i[6]} maxq {i[7]} | recorded {len(seq)}: " + " ".join(seq))
print("all recorded iterations by kind (+ full step / - partial step with a drop):", dict(tot))
