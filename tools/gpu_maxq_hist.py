"""development (round 4): distribution of the LARGEST working set an agent passes through (info[7]) -- what a small first T tier
with an extension pool has to cover.  usage: python tools/gpu_maxq_hist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench

def report(tag, out):
    inf = out["info"].reshape(-1, 8); st = out["status"].ravel()
    mq, it = inf[:, 7], inf[:, 4]
    tot = max(int(it.sum()), 1)
    print(f"{tag}: agents {st.size} mean iters {it.mean():.2f} maxq max {mq.max()}")
    for lo, hi in ((0, 1), (1, 9), (9, 17), (17, 25), (25, 33), (33, 41), (41, 49), (49, 65)):
        m = (mq >= lo) & (mq < hi)
        print(f"   maxq {lo:2d}-{hi - 1:2d}: {m.sum():6d} agents ({100.0 * m.mean():5.2f} %), iterations {100.0 * it[m].sum() / tot:5.1f} %, infeasible {((st[m] & 8) != 0).sum()}")

cfg, N, S = wl.CONFIGS["C2"], 100, 512
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
d = mp.Dmpc("hard", **kw)
l, _, _ = d.init_batch(po, pf)
z = np.zeros_like(po)
report("hard C2 step 2", d.step_batch(l, po, z, z, pf))
for variant in ("bound", "ondemand"):
    cfgv = dict(cfg, variant=variant)
    dv = mp.Dmpc(variant, **wl.solver_kwargs(cfgv, N))
    l2, xp, xv, xa, pf2, alive = bench.capture_state(dv, cfgv, S, N, 12, wl.SEED0 + 2)
    report(f"{variant} C2 replay step 12", dv.step_batch(l2, xp, xv, xa, pf2))
cfg4, N4 = dict(wl.CONFIGS["C4"]), 10000
d4 = mp.Dmpc("bound", **wl.solver_kwargs(cfg4, N4))
po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)
l4, _, _ = d4.init_batch(po4, pf4)
xp, xv, xa = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
for k in range(4):
    o = d4.step_batch(l4, xp, xv, xa, pf4)
    report(f"C4 bound N=1e4 step {k + 2}", o)
    ok = o["status"] == 1
    l4 = np.where(ok[..., None], o["p"], l4); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
