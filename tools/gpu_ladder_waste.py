"""development (round 4): how many solver tries of the retry-ladder agents are wasted -- tries the solve kernel starts (crash start, 8 iterations,
certificate) although a later level is the first feasible one?  tries (info[2]) against the scan's ladder start (hdr[6], levels skipped by the
per-row test), for the 10^4-agent scene and the bound replay."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
L = _lib.load(); L.dmpc_debug_read_hdr.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
def report(tag, d, out, T):
    hdr = np.zeros((T, 8), dtype=np.int32); assert L.dmpc_debug_read_hdr(d._ctx, hdr.ctypes.data_as(C.c_void_p), T) == 0
    inf = out["info"].reshape(-1, 8); tries, it = inf[:, 2], inf[:, 4]; ls = hdr[:, 6]
    lad = tries > 1
    started = tries - ls           # levels the solve kernel touched (incl. those its own certificate loop skipped)
    top = np.argsort(it)[-8:][::-1]
    print(f"{tag}: agents with tries > 1: {lad.sum()}; of them scan start >= 1: {(lad & (ls >= 1)).sum()}; tries - scan start: mean {started[lad].mean() if lad.any() else 0:.2f} max {started.max()}")
    print("    longest: " + "  ".join(f"[it {it[a]} tries {tries[a]} scan-start {ls[a]}]" for a in top))
cfg4, N4 = dict(wl.CONFIGS["C4"]), 10000
d4 = mp.Dmpc("bound", **wl.solver_kwargs(cfg4, N4))
po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)
l4, _, _ = d4.init_batch(po4, pf4)
xp, xv, xa = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
for k in range(5):
    o = d4.step_batch(l4, xp, xv, xa, pf4)
    report(f"C4 step {k + 2}", d4, o, N4)
    ok = o["status"] == 1
    l4 = np.where(ok[..., None], o["p"], l4); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
cfg, N, S = dict(wl.CONFIGS["C2"], variant="bound"), 100, 512
dv = mp.Dmpc("bound", **wl.solver_kwargs(cfg, N))
l2, xp, xv, xa, pf2, alive = bench.capture_state(dv, cfg, S, N, 12, wl.SEED0 + 2)
report("bound replay step 12", dv, dv.step_batch(l2, xp, xv, xa, pf2), S * N)
