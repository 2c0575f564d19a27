"""development (round 4): C3 (softall, N = 1000) and C5 (repair, N = 200) batches: scan / solve split per step, rows kept per agent, iterations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl, _lib
import ctypes as C
L = _lib.load(); L.dmpc_debug_read_hdr.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for cname, variant, Nc, Sc in (("C3", "softall", 1000, 16), ("C5", "repair", 200, 64)):
    cfgc = dict(wl.CONFIGS[cname]); kwc = wl.solver_kwargs(cfgc, Nc)
    dc = mp.Dmpc(variant, **kwc)
    poc, pfc = wl.make_scenes_device(dc, cfgc, Sc, Nc, wl.SEED0 + 7)
    lc, _, _ = dc.init_batch(poc, pfc)
    xp_, xv_, xa_ = poc.copy(), np.zeros_like(poc), np.zeros_like(poc)
    for k in range(5):
        dc.profile(True)
        oc = dc.step_batch(lc, xp_, xv_, xa_, pfc)
        sms, cms, _ = dc.profile_read2()
        T = Sc * Nc
        hdr = np.zeros((T, 8), dtype=np.int32); L.dmpc_debug_read_hdr(dc._ctx, hdr.ctypes.data_as(C.c_void_p), T)
        inf = oc["info"].reshape(-1, 8)
        print(f"{cname} {variant} step {k+2}: scan {cms*1e3:.0f} us solve {sms*1e3:.0f} us | rows kept mean {hdr[:,0].mean():.0f} max {hdr[:,0].max()} (reference rows {inf[:,1].mean():.0f}) | iterations mean {inf[:,4].mean():.1f} max {inf[:,4].max()} | maxq mean {inf[:,7].mean():.1f} max {inf[:,7].max()}")
        okm = (oc["status"] == 1)
        lc = np.where(okm[..., None], oc["p"], lc); xp_ = np.where(okm[..., None], oc["p"][..., :3], xp_)
        xv_ = np.where(okm[..., None], oc["v"][..., :3], xv_); xa_ = np.where(okm[..., None], oc["a"][..., :3], xa_)
