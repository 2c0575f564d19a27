"""development (round 4): the fp32-factor sweep of tests/test_gpu_precision.py over the dependence threshold (option f32_dep_exp) and all variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from helpers import init_table
CASES_ALL = [("C2", "hard", 100, {}), ("C4", "bound", 100, {}), ("C4", "bound2", 100, {}), ("C4", "all3", 100, {}), ("C5", "repair", 200, {"term": -1e7}), ("C3", "softall", 200, {}),
         ("C2", "ondemand", 100, {}), ("C2", "ellip", 100, {}), ("C4", "cpp", 100, {"term": -1e6}), ("C4", "cpp1", 100, {})]
CASES = [c for c in CASES_ALL if c[1] in os.environ.get("ONLY", c[1])]
exps = [int(x) for x in sys.argv[1:]] or [9]
for cfgname, variant, N, over in CASES:
    cfg = wl.CONFIGS[cfgname]; kw = dict(wl.solver_kwargs(cfg, N), **over); S = 8
    po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 5)
    d64 = mp.Dmpc(variant, **kw)
    ds = {e: mp.Dmpc(variant, precision="f32factor", **kw).debug_option("f32_dep_exp", e) for e in exps}
    l = np.stack([init_table(po[s], pf[s]) for s in range(S)])
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    acc = {e: dict(tot=0, same=0, tries=0, bad=0, worst=0.0, it64=0, it32=0) for e in exps}
    for k in range(8):
        a = d64.step_batch(l, xp, xv, xa, pf)
        for e in exps:
            b = ds[e].step_batch(l, xp, xv, xa, pf); r = acc[e]
            eq = a["status"] == b["status"]; r["tot"] += eq.size; r["same"] += int(eq.sum())
            te = eq & (a["info"][..., 2] == b["info"][..., 2]); r["tries"] += int((eq & ~te).sum())
            r["bad"] += int(((b["status"] & 48) != 0).sum())
            ok = te & ((a["status"] & 1) == 1)
            if ok.any(): r["worst"] = max(r["worst"], float(np.abs(a["p"][ok] - b["p"][ok]).max()))
            r["it64"] += int(a["info"][..., 4].sum()); r["it32"] += int(b["info"][..., 4].sum())
        upd = (a["status"] & 1) == 1
        l = np.where(upd[..., None], a["p"], l); xp = np.where(upd[..., None], a["p"][..., :3], xp)
        xv = np.where(upd[..., None], a["v"][..., :3], xv); xa = np.where(upd[..., None], a["a"][..., :3], xa)
    for e in exps:
        r = acc[e]
        print(f"{cfgname} {variant:8s} dep 1e-{e}: {r['tot']} agent-steps, status differs {r['tot'] - r['same']}, same status but other retry count {r['tries']}, capacity/itercap {r['bad']}, "
              f"l_inf(p) same status+tries {r['worst']:.2e}, iterations fp64 {r['it64']} fp32-factor {r['it32']}")
