#!/usr/bin/env python3
"""Development probe of the reduced solver (csrc/dmpc_rsolve.hip): one teacher-forced MPC step of a scene by the reduced solver, by the
general solver (debug option reduced_solver = 0) and by the oracle; prints the agents where they part.
usage: gpu_rsolve_check.py [golden | c4:N:steps:seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiagent_planning_amd as mp  # noqa: E402
from multiagent_planning_amd import workload as wl  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import load_golden, step14_inputs  # noqa: E402


def compare(tag, kw, l, xp, xv, xa, pf, variant="bound"):
    red = mp.Dmpc(variant, device=0, **kw)
    gen = mp.Dmpc(variant, device=0, **kw)
    import ctypes as C
    from multiagent_planning_amd import _lib
    L = _lib.load(); L.dmpc_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dmpc_debug_trace(red._ctx, -7, 8, None)
    gen.debug_option("reduced_solver", 0)
    o_r = red.step_batch(l, xp, xv, xa, pf)
    o_g = gen.step_batch(l, xp, xv, xa, pf)
    ref = orc.step(orc.make_params(variant, **kw), l, xp, xv, xa, pf, nthreads=os.cpu_count())
    hb = np.zeros((8, 8)); L.dmpc_debug_trace(red._ctx, -7, 8, hb.ctypes.data_as(C.c_void_p))
    print("      give-up reasons (1 rows>64, 2 steps, 3 iter cap, 4 inner cap, 5 third wall, 6 nh>5, 7 bad pivot, 8 crash singular, 9 farkas, 10 zero steps, 11 cycle):", hb.view(np.int32).ravel()[:12].tolist(), "agents", hb.ravel()[16:28].tolist())
    st_r, st_g, st_o = o_r["status"].ravel(), o_g["status"].ravel(), ref["status"].ravel()
    ir, ig, io = o_r["info"].reshape(-1, 8), o_g["info"].reshape(-1, 8), ref["info"].reshape(-1, 8)
    N = len(st_o)
    bad = []
    worst = 0.0
    for n in range(N):
        e = float(np.abs(o_r["a"].reshape(N, 45)[n] - ref["a"][n]).max()) if (st_o[n] & 1) and (st_r[n] & 1) else 0.0
        eg = float(np.abs(o_g["a"].reshape(N, 45)[n] - ref["a"][n]).max()) if (st_o[n] & 1) and (st_g[n] & 1) else 0.0
        ep = float(max(np.abs(o_r["p"].reshape(N, 45)[n] - ref["p"][n]).max(), np.abs(o_r["v"].reshape(N, 45)[n] - ref["v"][n]).max())) if (st_o[n] & 1) and (st_r[n] & 1) else 0.0
        worst = max(worst, e, ep)
        if st_r[n] != st_o[n] or ir[n, 2] != io[n, 2] or e > 1e-8 or ep > 1e-8:
            bad.append((n, int(st_r[n]), int(st_o[n]), int(ir[n, 2]), int(io[n, 2]), e, ep, eg, int(ir[n, 1]), int(ir[n, 4]), int(ir[n, 7]), int(ig[n, 4])))
    print(f"{tag}: N {N}  worst l_inf(a, p) reduced vs oracle {worst:.2e}  disagreeing agents {len(bad)}  mean EQPs {ir[:, 4].mean():.2f}  general iters {ig[:, 4].mean():.2f}")
    same = int(((ir[:, 4] == ig[:, 4]) & (ir[:, 7] == ig[:, 7]) & (ig[:, 4] > 3)).sum())
    top = np.argsort(-ir[:, 4])[:8]
    print("      agents that look like the general solver's (flagged):", same, " longest (agent, EQPs, rows, tries):", [(int(n), int(ir[n, 4]), int(ir[n, 1]), int(ir[n, 2])) for n in top])
    for b in bad[:25]:
        print("   agent %d status %d/%d tries %d/%d err a %.2e p %.2e (general %.2e) rows %d eqps %d maxq %d gen.iters %d" % b)
    return len(bad)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "golden"
    if what.startswith("trans"):   # whole transitions, reduced against general solver
        S = int(what.split(":")[1]) if ":" in what else 4
        cfg = wl.CONFIGS["C4"]; N = 100
        kw = wl.solver_kwargs(cfg, N)
        po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 55)
        red = mp.Dmpc("bound", **kw); gen = mp.Dmpc("bound", **kw); gen.debug_option("reduced_solver", 0)
        a, b = red.transition(po, pf, 151, cfg["error_tol"]), gen.transition(po, pf, 151, cfg["error_tol"])
        print("scene status reduced", a["scene_status"], "general", b["scene_status"])
        print("K_T used reduced", a["K_T_used"], "general", b["K_T_used"])
        for key in ("pk", "vk", "ak"):
            print(key, "max diff", float(np.abs(a[key] - b[key]).max()))
        return
    if what.startswith("batch"):   # S scenes of 100 agents, closed loop by the general solver, every step compared
        S = int(what.split(":")[1])
        cfg = wl.CONFIGS["C4"]; N = 100
        kw = wl.solver_kwargs(cfg, N)
        po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 55)
        red = mp.Dmpc("bound", **kw); gen = mp.Dmpc("bound", **kw); gen.debug_option("reduced_solver", 0)
        l, _, _ = gen.init_batch(po, pf)
        xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
        for k in range(2, 12):
            o_r, o_g = red.step_batch(l, xp, xv, xa, pf), gen.step_batch(l, xp, xv, xa, pf)
            ok = (o_g["status"] & 1) == 1
            dst = (o_r["status"] != o_g["status"]).sum()
            e = max(float(np.abs(o_r[key][ok] - o_g[key][ok]).max()) for key in ("p", "v", "a"))
            print(f"step {k}: status differences {dst} of {ok.size}  max diff on solved agents {e:.2e}  statuses reduced {np.unique(o_r['status'], return_counts=True)}")
            l = np.where(ok[..., None], o_g["p"], l)
            xp = np.where(ok[..., None], o_g["p"][..., :3], xp); xv = np.where(ok[..., None], o_g["v"][..., :3], xv); xa = np.where(ok[..., None], o_g["a"][..., :3], xa)
        return
    if what == "golden":
        g, kw = load_golden("failure_rate2_bound")
        compare("failure_rate2 step 14", kw, *step14_inputs(g))
        g, kw = load_golden("comp_kctr_3_bound2")
        l, xp, xv, xa, pf = g["l"], g["pk"][:, 12], g["vk"][:, 12], g["ak"][:, 12], g["pf"]
        compare("comp_kctr_3 step 14 (as bound)", kw, l, xp, xv, xa, pf)
        return
    _, N, steps, seed = what.split(":")
    N, steps, seed = int(N), int(steps), int(seed)
    cfg = dict(wl.CONFIGS["C4"]); cfg["N"] = N
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 1, N, seed)
    po, pf = po[0], pf[0]
    prm = orc.make_params("bound", **kw)
    l = np.stack([orc.init_one(po[n], pf[n], cfg["h"], 15)[0] for n in range(N)])
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    for step in range(2, 2 + steps):
        compare(f"C4-like N {N} step {step}", kw, l, xp, xv, xa, pf)
        ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=os.cpu_count())
        ok = (ref["status"] & 1) == 1
        l = np.where(ok[:, None], ref["p"], l)
        xp = np.where(ok[:, None], ref["p"][:, :3], xp)
        xv = np.where(ok[:, None], ref["v"][:, :3], xv)
        xa = np.where(ok[:, None], ref["a"][:, :3], xa)


if __name__ == "__main__":
    main()
