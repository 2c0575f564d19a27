"""development (DEV_TRACE build: python tools/with_trace_lib.py tools/gpu_order_ideal.py [variant]): the headline launch in the built-in launch order against the
order by the agents' MEASURED solve durations (the bound of any heaviness predictor) and by their true iteration counts."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "hard"
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2) if variant != "hard" else (None,) * 6
if variant == "hard":
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
    l, _, _ = d.init_batch(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
L = _lib.load()
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.dmpc_debug_set_order.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
T = S * N
cap = T * 2 // 8 + 8
def run(name, order, trace=False):
    if order is None: L.dmpc_debug_set_order(d._ctx, None, 0)
    else:
        o = np.ascontiguousarray(order, dtype=np.int32)
        L.dmpc_debug_set_order(d._ctx, o.ctypes.data_as(C.POINTER(C.c_int)), T)
    if trace: assert L.dmpc_debug_trace(d._ctx, -5, cap, None) == 0
    else: L.dmpc_debug_trace(d._ctx, -1, 0, None)
    for _ in range(2): out = d.step_batch(l, xp, xv, xa, pf)
    d.profile(True)
    for _ in range(6): out = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    d.profile(False)
    print(f"{name:40s} solve {sms*1e3:7.1f} us  scan+order {cms*1e3:6.1f} us")
    if trace:
        buf = np.zeros(cap * 8)
        assert L.dmpc_debug_trace(d._ctx, -5, cap, buf.ctypes.data_as(C.c_void_p)) == 0
        return out, buf[:T * 2].reshape(T, 2)[:, 1] * 1e-2
    return out, None
out, dur = run("built-in order (traced)", None, trace=True)
it = out["info"].reshape(-1, 8)[:, 4]
print(f"sum of durations / 2304 waves = {dur.sum()/2304:.0f} us; longest {dur.max():.0f} us")
run("built-in order", None)
run("by measured duration (ideal)", np.argsort(-dur, kind="stable"))
run("by true iteration count", np.argsort(-it, kind="stable"))
run("agent index (no order)", np.arange(T))
run("built-in order again", None)
