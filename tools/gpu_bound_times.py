"""development (library built with `make DEV_TRACE=1`): the bench's solveSoftDMPCbound replay (512 scenes at MPC step 12): per-position
solve durations, wave end times and the iteration histogram -- is that launch bound by its bulk or by its tail?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
cfg, N, S = wl.CONFIGS["C2"], 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("bound", **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
cap = S * N * 2 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -3, cap, None) == 0
    d.profile(True)
    out = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -3, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:S * N * 2].reshape(S * N, 2)
start = (t[:, 0] - t[:, 0].min()) * 1e-2; dur = t[:, 1] * 1e-2
it = out["info"][..., 4].ravel()
print(f"solve {sms*1e3:.0f} us scan {cms*1e3:.0f} us; durations mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; sum/2048 = {dur.sum()/2048:.0f} us; last end {np.max(start+dur):.0f}")
print("iterations histogram:", {k: int(((it >= a) & (it < b)).sum()) for k, (a, b) in {"0": (0, 1), "1": (1, 2), "2-3": (2, 4), "4-7": (4, 8), "8-15": (8, 16), "16-31": (16, 32), "32+": (32, 10000)}.items()})
for lo, hi in ((0, 2048), (2048, 4096), (4096, 16384), (16384, 51200)):
    dd = dur[lo:hi]; print(f"positions {lo}-{hi}: mean {dd.mean():.1f} us p99 {np.percentile(dd,99):.0f} max {dd.max():.0f}")
late = np.argsort(start + dur)[-6:]
print("last to end: position", late, "start", np.round(start[late]), "dur", np.round(dur[late]))
inf = out["info"].reshape(-1, 8)
heavy = it >= 32
print("agents with >= 32 iterations:", heavy.sum(), "| tries among them:", np.bincount(inf[heavy, 2]), "| rows among them: mean", inf[heavy, 1].mean(), "min", inf[heavy, 1].min())
print("all agents: tries >= 2:", (inf[:, 2] >= 2).sum(), " of which heavy:", (heavy & (inf[:, 2] >= 2)).sum(), "; rows >= 15:", (inf[:, 1] >= 15).sum(), "of which heavy", (heavy & (inf[:, 1] >= 15)).sum(),
      "; rows >= 20:", (inf[:, 1] >= 20).sum(), "of which heavy", (heavy & (inf[:, 1] >= 20)).sum())
for thr in (8, 16, 32, 64):
    h = it >= thr
    print(f"  it >= {thr}: {h.sum()} agents; with viol_k>0: {(h & (inf[:,0]>0)).sum()}; mean rows {inf[h,1].mean():.1f}; tries>=2: {(h & (inf[:,2]>=2)).sum()}")
print("agents with viol_k>0:", (inf[:, 0] > 0).sum(), "their mean iterations", it[inf[:, 0] > 0].mean(), "; without:", it[inf[:, 0] == 0].mean())
