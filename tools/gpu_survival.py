"""development aid: survival of scenes and iteration statistics per MPC step for a config."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = wl.CONFIGS[name]; N = int(sys.argv[4]) if len(sys.argv) > 4 else cfg["N"]
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(cfg["variant"], **kw)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 2)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
alive = np.ones(S, bool)
for k in range(1, steps):
    out = d.step_batch(l, xp, xv, xa, pf)
    ok = out["status"] == 1
    inf = out["info"]
    al = alive.copy()
    print(f"step {k+1:2d}: alive_in={al.sum():3d} solved={ok[al].mean() if al.any() else 0:.4f} infeas={((out['status'][al]&8)!=0).mean() if al.any() else 0:.4f} "
          f"iters mean={inf[al][...,4].mean() if al.any() else 0:.1f} max={inf[al][...,4].max() if al.any() else 0} rows mean={inf[al][...,1].mean() if al.any() else 0:.0f} maxq={inf[al][...,7].max() if al.any() else 0}")
    alive &= ok.all(axis=1)
    upd = alive[:, None] & ok
    l = np.where(upd[..., None], out["p"], l); xp = np.where(upd[..., None], out["p"][..., :3], xp)
    xv = np.where(upd[..., None], out["v"][..., :3], xv); xa = np.where(upd[..., None], out["a"][..., :3], xa)
