"""development probe (CPU, numpy model): how early would a dual-objective bound prove the infeasible hard-constrained
QPs of the C2 bench workload?  The dual active-set iterate x minimises the cost over the working set, so f(x) is a lower
bound of the constrained optimum; once it exceeds the maximum of f over the acceleration box no feasible point exists."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import structured_model as sm   # tests/dev/structured_model.py (numpy model used by these probes only)
from oracle import oracle as orc
from multiagent_planning_amd import workload as wl

cfg = wl.CONFIGS["C2"]; N = 100; K = 15
kw = wl.solver_kwargs(cfg, N)
nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
po_all, pf_all = wl.make_scenes(cfg, nsc)
h = kw["h"]
def pivot(c, v, Tm):   # H-norm pivot rule of the HIP kernel
    if c.typ in (sm.BOXHI, sm.BOXLO): return v / np.sqrt(Tm.Hinv1[c.ka, c.ka])
    if c.typ in (sm.POSHI, sm.POSLO): return v / np.sqrt(Tm.P1[c.ky, c.ky])
    if c.typ == sm.COLL: return v / np.sqrt(Tm.P1[c.ky, c.ky] * (c.yv @ c.yv) + (0.5 * c.ss * c.ss if c.si >= 0 else 0.0))
    return v * np.sqrt(2.0)
sm.PIVOT = pivot
res = []
orig = sm.solve_structured
for s in range(nsc):
    po, pf = po_all[s], pf_all[s]
    l = np.stack([np.asarray(orc.init_one(po[i], pf[i], h, K)[0]).reshape(-1) for i in range(N)])
    for n in range(N):
        trace = []; Tb = {}
        def hook(a, f, W, iters):
            H1 = Tb["H1"]
            g = H1 @ a + f   # gradient, = -(N lambda) by stationarity
            fk = (g * a).sum() - kw["alim"] * np.abs(g).sum()   # > 0: the multipliers are a Farkas certificate against the box
            trace.append((iters, sum(0.5 * a[:, ax] @ H1 @ a[:, ax] + f[:, ax] @ a[:, ax] for ax in range(3)), len(W), fk))
            Tb["f"] = f
        def wrapped(T, *a, **k):
            Tb["T"] = T; Tb["H1"] = np.linalg.inv(T.Hinv1)
            return orig(T, *a, **k)
        sm.ITER_HOOK = hook; sm.solve_structured = wrapped
        out = sm.solve_agent_model("hard", K, h, kw["rmin"], cfg["c"], kw["alim"], kw["Q1"], kw["S1"], -5e4, kw["pmin"], kw["pmax"], l, n, po[n], np.zeros(3), np.zeros(3), pf[n], [])
        sm.solve_structured = orig
        if out["status"] != 8 or not trace: continue
        H1 = Tb["H1"]; f = Tb["f"]; alim = kw["alim"]
        lam_max = np.linalg.eigvalsh(H1)[-1]
        cheap = sum(0.5 * lam_max * K * alim ** 2 + alim * np.abs(f[:, ax]).sum() for ax in range(3))
        # tighter: vertex of the box in the direction of the gradient sign (a lower bound of the max) and the
        # diagonal-dominance bound  max <= sum_i (0.5*rowsum|H|_i*alim^2 + |f_i| alim)
        rowsum = np.abs(H1).sum(axis=1)
        gersh = sum((0.5 * rowsum * alim ** 2).sum() + alim * np.abs(f[:, ax]).sum() for ax in range(3))
        its = trace[-1][0]
        hit_c = next((it for it, v, q, fk in trace if fk > 0), None)
        hit_g = next((it for it, v, q, fk in trace if v > gersh), None)
        res.append((s, n, its, hit_g, hit_c, trace[-1][1], gersh, cheap))
print("infeasible agents:", len(res), "of", nsc * N)
tot = sum(r[2] for r in res); sav = sum((r[2] - r[3]) for r in res if r[3] is not None)
print("iterations spent on them:", tot, " saved by the (Gershgorin) bound:", sav, " by the Farkas test:", sum((r[2] - r[4]) for r in res if r[4] is not None), " agents it proves:", sum(r[3] is not None for r in res))
for r in sorted(res, key=lambda r: -r[2])[:25]:
    print("scene %d agent %3d iterations %4d  bound hit at %s (Farkas %s)  final dual value %.3e  bounds %.3e %.3e" % r)
