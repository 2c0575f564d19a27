"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_key_features.py): per agent of the C4 closed loop (N = 10^4) the scan's
launch-order key with its raw ingredients (feature bits of the DEV_TRACE scan) next to what the solve then cost (iterations, tries, work estimate).
Saved to gpurun_out/key_features.npz for tools/key_fit.py (CPU): which key orders the queue best?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
N = 10000
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc(cfg["variant"], **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
assert L.dmpc_debug_trace(d._ctx, -6, 8, None) == 0
infos = []
for k in range(9):
    out = d.step_batch(l, xp, xv, xa, pf)
    infos.append(out["info"][0].copy())
    st = out["status"][0]; ok = st == 1
    l = np.where(ok[None, :, None], out["p"], l); xp = np.where(ok[None, :, None], out["p"][..., :3], xp)
    xv = np.where(ok[None, :, None], out["v"][..., :3], xv); xa = np.where(ok[None, :, None], out["a"][..., :3], xa)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/key_features.npz", info=np.array(infos))
i = infos[3]
print("step 5: cost mean", i[:, 3].mean(), "iters mean", i[:, 4].mean(), "key word sample", [hex(x) for x in i[:5, 5]])
