"""development: iteration / retry statistics of the first MPC steps of C4 (N = 10^4, one scene)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = wl.CONFIGS["C4"]; kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d = mp.Dmpc(cfg["variant"], **kw)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(1, int(os.environ.get('STEPS', '9'))):
    d.profile(True)
    out = d.step_batch(l, xp, xv, xa, pf)
    solve_ms, scan_ms, _ = d.profile_read2()
    st = out["status"][0]; inf = out["info"][0]
    it, tr, na, nr = inf[:, 4], inf[:, 2], inf[:, 6], inf[:, 7]
    print(f"step {k+1}: scan {scan_ms:.3f} solve {solve_ms:.3f} ms solved {(st==1).mean():.4f} iters mean {it.mean():.1f} p50 {np.median(it):.0f} p90 {np.percentile(it,90):.0f} p99 {np.percentile(it,99):.0f} max {it.max()}"
          f" | with rows: {(nr>0).mean():.3f}, their mean iters {it[nr>0].mean() if (nr>0).any() else 0:.1f}; without rows mean iters {it[nr==0].mean():.1f} nactive {na[nr==0].mean():.1f}"
          f" | tries>1: {(tr>1).sum()} agents, iters of those mean {it[tr>1].mean() if (tr>1).any() else 0:.0f} max tries {tr.max()}; maxq p50 {np.median(nr):.0f} p90 {np.percentile(nr,90):.0f} p99 {np.percentile(nr,99):.0f} max {nr.max()} (>32: {(nr>32).mean():.3f}, >48: {(nr>48).mean():.4f})")
    if os.environ.get("SURV"):
        P = l[0].reshape(N, 15, 3); lo = P.min(1); hi = P.max(1)
        R = 3 * kw["rmin"]; infl = np.array([R, R, R * kw["c"]])
        cnt = np.zeros(N, int); cnt3 = np.zeros(N, int)
        slo = P.reshape(N, 3, 5, 3).min(2); shi = P.reshape(N, 3, 5, 3).max(2)
        for i0 in range(0, N, 500):
            ov = ((lo[None] <= hi[i0:i0+500, None, :] + infl) & (hi[None] >= lo[i0:i0+500, None, :] - infl)).all(-1)
            cnt[i0:i0+500] = ov.sum(1) - 1
            ov3 = ((slo[None] <= shi[i0:i0+500, None] + infl) & (shi[None] >= slo[i0:i0+500, None] - infl)).all(-1).any(-1)
            cnt3[i0:i0+500] = ov3.sum(1) - 1
        print(f"   table of this step: segment survivors mean {cnt3.mean():.0f} p90 {np.percentile(cnt3, 90):.0f} max {cnt3.max()}; whole-box survivors mean {cnt.mean():.0f} p90 {np.percentile(cnt, 90):.0f} max {cnt.max()}; extent mean {(hi-lo).mean(0).round(2)}; rows built mean {inf[:,1].mean():.1f} max {inf[:,1].max()}; agents with violation {(inf[:,0]>0).mean():.3f}")
    ok = st == 1
    l = np.where(ok[None, :, None], out["p"], l); xp = np.where(ok[None, :, None], out["p"][..., :3], xp)
    xv = np.where(ok[None, :, None], out["v"][..., :3], xv); xa = np.where(ok[None, :, None], out["a"][..., :3], xa)
