"""development aid: run the first MPC solves of the BASELINE configs C3/C4/C5 on one GPU and report
status / timing (functional check of large-N paths)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
for name, N, S, steps in [("C5", 200, 4, 6), ("C3", 1000, 1, 4), ("C4", 10000, 1, 4)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    cfg = wl.CONFIGS[name]
    kw = wl.solver_kwargs(cfg, N)
    t0 = time.time()
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + int(name[1]))
    tg = time.time() - t0
    d = mp.Dmpc(cfg["variant"], **kw)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(1, steps):
        d.profile(True)
        out = d.step_batch(l, xp, xv, xa, pf)
        solve_ms, scan_ms, _ = d.profile_read2()
        st = out["status"]; inf = out["info"]
        print(f"{name} N={N} S={S} step {k+1}: solved={(st==1).mean():.4f} infeas={((st&8)!=0).mean():.4f} coll={((st&4)!=0).mean():.4f} "
              f"invalid={((st&48)!=0).sum()} iters mean/max={inf[...,4].mean():.1f}/{inf[...,4].max()} rows mean/max={inf[...,1].mean():.0f}/{inf[...,1].max()} "
              f"maxq={inf[...,7].max()} scan={scan_ms:.3f} ms solve={solve_ms:.3f} ms (scene gen {tg:.1f}s)")
        ok = st == 1
        l = np.where(ok[..., None], out["p"], l); xp = np.where(ok[..., None], out["p"][..., :3], xp)
        xv = np.where(ok[..., None], out["v"][..., :3], xv); xa = np.where(ok[..., None], out["a"][..., :3], xa)
