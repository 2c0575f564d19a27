"""development (round 5): warm start of closed loops -- the same closed loop driven through step_batch cold and warm (dmpc_warm_start): agreement of
the outputs / statuses / ladder counts, iterations, device time per step.   usage: python tools/gpu_warm_probe.py [C4|C3|C5|C2b] [steps] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
which = sys.argv[1] if len(sys.argv) > 1 else "C4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
cfgname, N, S, variant = {"C4": ("C4", 10000, 1, "bound"), "C3": ("C3", 1000, 16, "softall"), "C5": ("C5", 200, 64, "repair"),
                          "C2b": ("C4", 100, 512, "bound"), "all3": ("C4", 100, 128, "all3"), "bound2": ("C4", 100, 128, "bound2")}[which]
if len(sys.argv) > 3: S = int(sys.argv[3])
cfg = dict(wl.CONFIGS[cfgname]); kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 4)
def run(warm):
    d = mp.Dmpc(variant, **kw)
    if warm: d.warm_start(True)
    d.profile(True)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    outs = []
    for k in range(steps):
        o = d.step_batch(l, xp, xv, xa, pf)
        sv, sc, n = d.profile_read2()
        o["solve_ms"], o["scan_ms"] = sv, sc
        outs.append(o)
        ok = (o["status"] & 1) == 1
        l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
        xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
    return outs
cold, warm = run(False), run(True)
for k, (c, w) in enumerate(zip(cold, warm)):
    ok = (c["status"] & 1) == 1
    dp = np.abs(c["p"] - w["p"])[ok].max() if ok.any() else 0.0
    da = np.abs(c["a"] - w["a"])[ok].max() if ok.any() else 0.0
    ic, iw = c["info"][..., 4], w["info"][..., 4]
    print(f"step {k + 2}: status== {np.array_equal(c['status'], w['status'])} branch== {np.array_equal(c['info'][..., :4], w['info'][..., :4])} "
          f"|dp| {dp:.2e} |da| {da:.2e}  iters cold mean {ic.mean():.1f} max {ic.max()}  warm mean {iw.mean():.1f} max {iw.max()}  "
          f"solve ms cold {c['solve_ms']:.3f} warm {w['solve_ms']:.3f}  scan {c['scan_ms']:.3f}/{w['scan_ms']:.3f}")
    if not np.array_equal(c["status"], w["status"]):
        bad = np.argwhere(c["status"] != w["status"])[:5]
        for b in bad: print("    status differs at", tuple(b), c["status"][tuple(b)], w["status"][tuple(b)], "info", c["info"][tuple(b)], w["info"][tuple(b)])
    elif not np.array_equal(c["info"][..., :4], w["info"][..., :4]):
        bad = np.argwhere((c["info"][..., :4] != w["info"][..., :4]).any(-1))[:5]
        for b in bad: print("    branch differs at", tuple(b), c["info"][tuple(b)], w["info"][tuple(b)])
