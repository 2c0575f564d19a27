"""development (library built with `make DEV_TRACE=1`; run with DMPC_DEBUG_OPTIONS=force_persist=1,tier1_qcap=64): duration of every agent's solve
at C4 N = 10^4 (third MPC step): is the launch bound by the sum of the work or by its longest agents?"""
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
cap = N * 2 // 8 + 8
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    assert L.dmpc_debug_trace(d._ctx, -3, cap, None) == 0
    d.profile(True)
    out = d.step_batch(l, xp, xv, xa, pf)
    solve_ms, scan_ms, _ = d.profile_read2()
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -3, cap, buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf[:N * 2].reshape(N, 2)
    start = (t[:, 0] - t[:, 0].min()) * 1e-2; dur = t[:, 1] * 1e-2
    it = out["info"][0, :, 4]
    print(f"step {k+2}: solve launch {solve_ms*1e3:.0f} us; durations mean {dur.mean():.1f} p50 {np.median(dur):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us; sum/1280 = {dur.sum()/1280:.0f} us; last end {np.max(start+dur):.0f} us; "
          f"iterations mean {it.mean():.1f} max {it.max()}; us per iteration (sum dur / sum it) {dur.sum()/it.sum():.2f}")
    late = np.argsort(start + dur)[-5:]
    print("   last to end: position", late, "start", np.round(start[late]), "dur", np.round(dur[late]))
    st = out["status"][0]; ok = st == 1
    l = np.where(ok[None, :, None], out["p"], l); xp = np.where(ok[None, :, None], out["p"][..., :3], xp)
    xv = np.where(ok[None, :, None], out["v"][..., :3], xv); xa = np.where(ok[None, :, None], out["a"][..., :3], xa)

# crash statistics (dmpc_debug_trace agent -4: info[0] = appended without a step, info[1] = crash rounds, info[3] = passes that dropped negative multipliers)
assert L.dmpc_debug_trace(d._ctx, -4, 8, None) == 0
out = d.step_batch(l, xp, xv, xa, pf)
inf = out["info"][0]
print(f"crash: appended mean {inf[:,0].mean():.1f} of {inf[:,4].mean():.1f} iterations (from the table {inf[:,5].mean():.1f}, by products {inf[:,6].mean():.1f}); rounds mean {inf[:,1].mean():.2f}; agents with a negative-multiplier drop {(inf[:,3]>0).mean():.3f}")
