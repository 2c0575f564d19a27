"""development aid (phase cycle counts need a library built with `make -C multiagent_planning_amd/csrc DEV_TIMERS=1`): dump the active-set iteration trace of one agent on the GPU."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multiagent_planning_amd as mp
from helpers import load_golden, step14_inputs
name, variant, agent = sys.argv[1], sys.argv[2], int(sys.argv[3])
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 120
g, kw = load_golden(name)
l, xp, xv, xa, pf = step14_inputs(g)
d = mp.Dmpc(variant, **kw)
L = d._L
L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.dmpc_debug_trace(d._ctx, agent, cap, None)
out = d.step_batch(l, xp, xv, xa, pf)
buf = np.zeros((cap, 8))
L.dmpc_debug_trace(d._ctx, agent, cap, buf.ctypes.data_as(C.c_void_p))
print("status", out["status"][agent], "info", out["info"][agent])
names = ["BH", "BL", "PH", "PL", "CO", "SU", "SL"]
for i, r in enumerate(buf[:-1]):
    if r[3] == 0 or i >= 6: break
    code = int(r[0])
    print(f"{i+1:4d} p={names[code>>16]}{code&0xffff:<4d} q={int(r[1]):2d} delta/spp={r[2]/r[3]:.3e} t1={r[4]:.4e} t2={r[5]:.4e} vp={r[6]:.3e} lam_p={r[7]:.3e}")

ph = buf[cap - 1]
print("phase cycles: scan+rows=%d setup=%d solve=%d out=%d | in solve: violation-scan=%d sdot+matvec=%d nu/delta=%d iters=%d" % tuple(int(x) for x in ph))
print("ratio/lambda=%d  step+append=%d  desc=%d  drop-path=%d" % tuple(buf[cap-2][:4]))
if ph[7] > 0: print("per iteration: solve=%.0f viol=%.0f matvec=%.0f nu=%.0f cycles" % (ph[2]/ph[7], ph[4]/ph[7], ph[5]/ph[7], ph[6]/ph[7]))
