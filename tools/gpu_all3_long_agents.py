"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_all3_long_agents.py [variant] [scenes]): the longest agents of a solveSoftDMPCall closed loop
(100 agents x S scenes, C2 box, MPC step 12): measured duration, iterations, ladder tries, working-set size -- what does an iteration of its heavy agents cost?"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "all3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = 100
cfg = dict(wl.CONFIGS["C2"], variant=variant); kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
if os.environ.get("PLAIN"):   # product build under the profiler: the step alone, five times
    for rep in range(5):
        d.profile(True); out = d.step_batch(l, xp, xv, xa, pf); sms, cms, _ = d.profile_read2()
    print(f"{variant} x {S} [{os.environ.get('DMPC_DEBUG_OPTIONS', '')}]: solve launches {sms*1e3:.0f} us, scan {cms*1e3:.0f} us; iterations max {out['info'][..., 4].max()}, most slots {out['info'][..., 7].max()}, statuses {np.unique(out['status'])}"); sys.exit(0)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
tot = S * N; cap = tot * 2 // 8 + 8
for rep in range(2):
    assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
    d.profile(True)
    out = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    buf = np.zeros(cap * 8)
    assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:tot * 2].reshape(tot, 2)
dur = t[:, 1] * 1e-2; st = (t[:, 0] - t[t[:, 1] > 0, 0].min()) * 1e-2
inf = out["info"].reshape(tot, 8)
ran = dur > 0
print(f"{variant} x {S} scenes: solve launch {sms*1e3:.0f} us; {ran.sum()} agents through the solver, duration mean {dur[ran].mean():.1f} us, sum / 1000 = {dur[ran].sum()/1000:.0f} us (divide by the thousands of wave slots of the launch: 2.048 at eight, 1.792 at seven, 3.072 at twelve waves per CU); longest {dur.max():.0f} us")
top = np.argsort(dur)[-10:][::-1]
for i in top:
    print(f"   agent {i}: start {st[i]:.0f} us, {dur[i]:.0f} us, iterations {inf[i,4]}, tries {inf[i,2]}, rows {inf[i,1]}, final slots {inf[i,6]}, most {inf[i,7]}: {dur[i]/max(inf[i,4],1):.2f} us per iteration")
for lo, hi in ((1, 10), (10, 30), (30, 80), (80, 160), (160, 2000)):
    m = ran & (inf[:, 4] >= lo) & (inf[:, 4] < hi)
    if m.any(): print(f"   iterations {lo:4d}-{hi:4d}: {m.sum():6d} agents, {dur[m].sum()/dur[ran].sum():.3f} of the work, {dur[m].sum()/inf[m,4].sum():.2f} us per iteration")
