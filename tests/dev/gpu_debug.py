"""ad-hoc GPU-vs-oracle comparison with diagnostics (development tool)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import multiagent_planning_amd as mp
from oracle import oracle as orc
from helpers import load_golden, oracle_params, step14_inputs, ALL_VARIANTS

names = ["failure_rate2_bound", "comp_kctr_3_bound2"]
variants = sys.argv[1:] or ALL_VARIANTS
for name in names:
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    for variant in variants:
        out = mp.Dmpc(variant, **kw).step_batch(l, xp, xv, xa, pf)
        ref = orc.step(oracle_params(variant, kw), l, xp, xv, xa, pf)
        st_o, st_r = out["status"], ref["status"]
        bad = np.where(st_o != st_r)[0]
        ok = ((st_r & 1) == 1) & ((st_o & 1) == 1)
        e = np.abs(out["p"][ok] - ref["p"][ok]).max() if ok.any() else 0
        ea = np.abs(out["a"][ok] - ref["a"][ok]).max() if ok.any() else 0
        io, ir = out["info"], ref["info"]
        print(f"{name:22s} {variant:9s} status_mismatch={len(bad):3d} linf(p)={e:.2e} linf(a)={ea:.2e} "
              f"iters mean/max={io[:,4].mean():.1f}/{io[:,4].max()} maxq={io[:,7].max()} nrows max={io[:,1].max()} "
              f"tries_mismatch={(io[:,2]!=ir[:,2]).sum()} rows_mismatch={(io[:,1]!=ir[:,7]).sum()} violk_mismatch={(io[:,0]!=ir[:,0]).sum()}")
        for n in bad[:6]:
            print("    agent", n, "gpu st", st_o[n], "info", io[n], "| oracle st", st_r[n], "info", ir[n])
