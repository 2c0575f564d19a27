"""development (round 4): campaign seed 43, scene 28, solveSoftDMPCall, MPC step 5: the agent whose retry count differs from the oracle's --
with and without the level check, and what the LP certificate (tests/certificates.py) says about the ladder levels in question"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, init_table
import certificates as cert
seed0, want_scene, want_variant = int(sys.argv[1]) if len(sys.argv) > 1 else 43, int(sys.argv[2]) if len(sys.argv) > 2 else 28, "all3"
rng = np.random.default_rng(seed0)
for it in range(want_scene + 1):
    N = int(rng.integers(2, 90))
    dense = rng.random() < 0.5
    cfgname = "C5" if dense else "C2"
    cfg = wl.CONFIGS[cfgname]; kw = wl.solver_kwargs(cfg, N)
    if rng.random() < 0.3:
        s = 0.8
        kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [s, s, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [s, s, 1])
    try:
        po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
    except Exception:
        continue
    po, pf = po[0], pf[0]
    for variant in ALL_VARIANTS:
        nst = int(rng.integers(2, 7))
        if it != want_scene or variant != want_variant: continue
        d = mp.Dmpc(variant, **kw); d0 = mp.Dmpc(variant, **kw).debug_option("no_level_check", 1); prm = orc.make_params(variant, **kw)
        l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
        for k in range(nst):
            out = d.step_batch(l, xp, xv, xa, pf); old = d0.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
            diff = np.where((out["status"] != ref["status"]) | (out["info"][:, 2] != ref["info"][:, 2]) | (old["info"][:, 2] != ref["info"][:, 2]))[0]
            for n in diff:
                print(f"scene {it} N={N} {cfgname} step {k+2} agent {n}: with level check status {out['status'][n]} tries {out['info'][n,2]} iters {out['info'][n,4]} | without: status {old['status'][n]} tries {old['info'][n,2]} iters {old['info'][n,4]} | oracle status {ref['status'][n]} tries {ref['info'][n,2]}")
                for lvl in range(max(int(min(out['info'][n,2], ref['info'][n,2])) - 2, 0), int(max(out['info'][n,2], ref['info'][n,2]))):
                    qp = orc.assemble_one(prm, l, int(n), xp[n], xv[n], xa[n], pf[n], level=lvl)
                    print(f"    ladder level {lvl} (try {lvl+1}): phase-1 LP optimum {cert.lp_infeasibility(qp):.3e}  (0 = feasible)")
            if len(diff) and "trace" in sys.argv:   # DEV_TRACE build: iterations of the try the GPU gives up on (max_tries = that try)
                import ctypes as C
                from multiagent_planning_amd import _lib
                L = _lib.load(); L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
                n = int(diff[0]); t_fail = int(ref["info"][n, 2])
                dt = mp.Dmpc(variant, max_tries=t_fail, **kw).debug_option("no_level_check", 1)
                cap = 600
                L.dmpc_debug_trace(dt._ctx, n, cap, None)
                o2 = dt.step_batch(l, xp, xv, xa, pf)
                buf = np.zeros((cap, 8)); L.dmpc_debug_trace(dt._ctx, n, cap, buf.ctypes.data_as(C.c_void_p))
                print(f"   max_tries {t_fail}: status {o2['status'][n]} info {o2['info'][n]}")
                rows = buf[buf[:, 3] != 0]
                big = np.where((np.abs(rows[:, 7]) > 1e7) | ((rows[:, 4] > 1e7) & np.isfinite(rows[:, 4]) & (rows[:, 5] == np.inf)))[0]
                first = int(big[0]) if len(big) else len(rows) - 14
                print(f"   {len(rows)} iterations recorded; first runaway at iteration {first + 1}")
                for r in rows[max(first - 12, 0):first + 4]:
                    pc = int(r[0]); print(f"     pivot type {pc >> 16} idx {pc & 0xffff}  q {int(r[1])}  delta {r[2]:.3e}  spp {r[3]:.3e}  t1 {r[4]:.3e}  t2 {r[5]:.3e}  viol {r[6]:.3e}  lam_p {r[7]:.3e}")
            okb = out["status"] & 1 == 1
            l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
            xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
