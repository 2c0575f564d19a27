"""development aid: run the step kernel a few times on a recorded scene (for rocprofv3 runs)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multiagent_planning_amd as mp
from helpers import load_golden, step14_inputs
variant = sys.argv[1] if len(sys.argv) > 1 else "hard"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g, kw = load_golden("failure_rate2_bound")
l, xp, xv, xa, pf = step14_inputs(g)
b = lambda a: np.ascontiguousarray(np.broadcast_to(a, (S,) + a.shape))
d = mp.Dmpc(variant, **kw)
for _ in range(reps):
    out = d.step_batch(b(l), b(xp), b(xv), b(xa), b(pf))
print("ok", (out["status"] & 1).mean())
