import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, init_table
want = {68, 120, 124, 150, 153, 159}
rng = np.random.default_rng(1)
for it in range(max(want) + 1):
    N = int(rng.integers(2, 90))
    cfgname = "C5" if rng.random() < 0.5 else "C2"
    cfg = wl.CONFIGS[cfgname]
    kw = wl.solver_kwargs(cfg, N)
    if rng.random() < 0.3:
        kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [0.8, 0.8, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [0.8, 0.8, 1])
    try:
        po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
    except Exception:
        continue
    po, pf = po[0], pf[0]
    for variant in ALL_VARIANTS:
        nst = int(rng.integers(2, 7))
        if it not in want or variant != "all3": continue
        d = mp.Dmpc(variant, **kw); prm = orc.make_params(variant, **kw)
        l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
        for k in range(nst):
            out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
            bad = np.where(out["status"] != ref["status"])[0]
            for n in bad:
                print(f"scene {it} N={N} step {k+2} agent {n}: gpu status {out['status'][n]} info {out['info'][n]} | oracle status {ref['status'][n]} info {ref['info'][n] if 'info' in ref else None}")
            okb = (out["status"] == ref["status"]) & ((ref["status"] & 1) == 1)
            if okb.any(): print(f"scene {it} step {k+2}: l_inf over agreeing solved agents {np.abs(out['p'][okb]-ref['p'][okb]).max():.2e}")
            ok = (ref["status"] == 1)[..., None]
            l = np.where(ok, ref["p"], l); xp = np.where(ok, ref["p"][..., :3], xp); xv = np.where(ok, ref["v"][..., :3], xv); xa = np.where(ok, ref["a"][..., :3], xa)
