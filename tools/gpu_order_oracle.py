"""development (round 4): how much is the launch ORDER worth on a workload?  One captured closed-loop state of a config is solved repeatedly, once with the
scan's key alone and once with option order_hint = 2 (the previous work estimate INSTEAD of the key) -- in a replay of ONE state that is the same step's own work, i.e. (nearly) the
perfect order, which no closed loop would see.  usage: python tools/gpu_order_oracle.py C3|C4|C5 [scenes] [mpc_step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(wl.CONFIGS[name]); N = cfg["N"]
S = int(sys.argv[2]) if len(sys.argv) > 2 else {"C3": 16, "C4": 1, "C5": 64}[name]
kstep = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, min(S, 4), N, wl.SEED0 + 9)
po, pf = np.tile(po, (S // min(S, 4), 1, 1)), np.tile(pf, (S // min(S, 4), 1, 1))
for hint in (0, 2):
    d = mp.Dmpc(cfg["variant"], **kw)
    d.debug_option("order_hint", hint)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for k in range(kstep - 2):
        o = d.step_batch(l, xp, xv, xa, pf)
        ok = (o["status"] == 1)[..., None]
        l = np.where(ok, o["p"], l); xp = np.where(ok, o["p"][..., :3], xp); xv = np.where(ok, o["v"][..., :3], xv); xa = np.where(ok, o["a"][..., :3], xa)
    for _ in range(3): o = d.step_batch(l, xp, xv, xa, pf)
    d.profile(True)
    for _ in range(8): o = d.step_batch(l, xp, xv, xa, pf)
    sms, cms, _ = d.profile_read2()
    it = o["info"][..., 4]
    print(f"{name} {cfg['variant']} {S} x {N} agents, MPC step {kstep}, order_hint {hint}: solve {sms * 1e3:7.1f} us, scan + lists + order {cms * 1e3:6.1f} us; iterations mean {it.mean():.1f} max {it.max()}")
