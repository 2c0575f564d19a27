"""development aid: randomized parity campaign -- many random scenes x all variants x several teacher-forced MPC steps,
GPU (C ABI) against the oracle; prints every mismatch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, init_table

nscen = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
bad = 0; total = 0; worst = 0.0
t0 = time.time()
for it in range(nscen):
    N = int(rng.integers(2, 90))
    dense = rng.random() < 0.5
    cfgname = "C5" if dense else "C2"
    cfg = wl.CONFIGS[cfgname]
    kw = wl.solver_kwargs(cfg, N)
    if rng.random() < 0.3:   # shrink the box: more conflicts, outbound cases
        s = 0.8
        kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [s, s, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [s, s, 1])
    try:
        po, pf = wl.make_scenes(dict(cfg), 1, N, int(rng.integers(1 << 30)))
    except Exception:
        continue
    po, pf = po[0], pf[0]
    for variant in ALL_VARIANTS:
        if os.environ.get("CAMPAIGN_ONLY") and variant not in os.environ["CAMPAIGN_ONLY"].split(","):
            rng.integers(2, 7)   # (keeps the random stream of the full campaign)
            continue
        d = mp.Dmpc(variant, **kw); prm = orc.make_params(variant, **kw)
        l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
        # slack multipliers scale with |term|: eps = -(t + sum lambda sigma)/2 loses |term| * 1e-16 * cond absolutely
        # (cpp1: DMPC::solveQP hard-codes a slack penalty of -1e6, ten times solveSoftDMPC's: ten times its allowance)
        tol = 1e-6 if variant == "cpp1" else (5e-8 if variant in ("softall", "repair", "softall_c") else 1e-9 * max(1.0, abs(kw["term"]) / 5e4))
        for k in range(int(rng.integers(2, 7))):
            out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
            total += N
            ok = (ref["status"] & 1) == 1
            msg = None
            if not np.array_equal(out["status"], ref["status"]): msg = f"status {out['status'][out['status']!=ref['status']][:5]} vs {ref['status'][out['status']!=ref['status']][:5]}"
            elif not (np.array_equal(out["info"][:, 0], ref["info"][:, 0]) and np.array_equal(out["info"][:, 1], ref["info"][:, 7]) and np.array_equal(out["info"][:, 2], ref["info"][:, 2])): msg = "branch record"
            else:
                e = max((np.abs(out[key][ok] - ref[key][ok]).max() if ok.any() else 0.0) for key in ("p", "v", "a"))
                worst = max(worst, e)
                if e > tol: msg = f"l_inf {e:.2e}"
            if msg:
                bad += 1
                dm = np.where((out['status'] != ref['status']) | (out['info'][:, 2] != ref['info'][:, 2]))[0][:3]
                msg += ' | agents ' + str([(int(a), int(out['status'][a]), int(ref['status'][a]), 'tries', int(out['info'][a, 2]), int(ref['info'][a, 2]), 'iters', int(out['info'][a, 4])) for a in dm])
                print(f"MISMATCH scene {it} N={N} {cfgname} variant {variant} step {k+2}: {msg}", flush=True)
            okb = out["status"] & 1 == 1
            l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
            xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
print(f"{nscen} scenes, {total} agent-steps compared, {bad} mismatching steps, worst l_inf {worst:.2e}, {time.time()-t0:.0f} s")
