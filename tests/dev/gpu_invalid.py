"""development aid: find agents flagged CAPACITY/ITERCAP in the emulated multi-GPU bench workload."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = wl.CONFIGS["C2"]; S, N = 64, 100 * G
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("hard", **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 2, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
st = out["status"]; inf = out["info"]
bad = np.argwhere((st & 48) != 0)
print("invalid agents:", len(bad), "of", st.size)
for s, n in bad[:10]:
    print("scene", s, "agent", n, "status", st[s, n], "info", inf[s, n])
print("maxq overall", inf[..., 7].max(), "max iters", inf[..., 4].max(), "max rows", inf[..., 1].max())
from oracle import oracle as orc
prm = orc.make_params("hard", **kw)
for s, n in [(22, 559), (24, 474), (27, 157), (46, 64), (49, 456), (0, 0), (1, 5)]:
    r = orc.solve_one(prm, l[s], n, xp[s, n], xv[s, n], xa[s, n], pf[s, n])
    e = abs(r["p"] - out["p"][s, n]).max() if r["status"] & 1 else 0.0
    print("scene", s, "agent", n, "gpu", st[s, n], inf[s, n][[1, 4, 7]], "oracle", r["status"], r["info"][[7, 4]], "linf(p)", e)
