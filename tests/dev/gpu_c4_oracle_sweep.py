"""development (not collected): the N = 10^4 closed loop (BASELINE configs[3], what bench.py's C4 line times) against the oracle on MANY agents --
every MPC step 2-10, the `per_step` agents with the most iterations, all agents with ladder retries, and a random sample, teacher forcing on the GPU's
own states; statuses, branch records and retry counts must be identical, trajectories within 1e-9.  The collected test
(tests/test_gpu_fullsize.py) does 48 agents at steps 3, 6, 10.   usage: python tests/dev/gpu_c4_oracle_sweep.py [per_step] [workers]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from multiprocessing import get_context
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc

per_step = int(sys.argv[1]) if len(sys.argv) > 1 else 400
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
CNAME = sys.argv[3] if len(sys.argv) > 3 else "C4"     # C4 (default) | C3 (solveSoftDMPC, 1 000 agents: a dense 999-row oracle QP per violating agent) | C5
cfg = wl.CONFIGS[CNAME]; N = cfg["N"]; VARIANT = cfg["variant"]
TOL = 1e-9 if CNAME == "C4" else 2e-8                 # (|term| = 1e5 / 1e6-scale multipliers: DESIGN section 6)
kw = wl.solver_kwargs(cfg, N)
G = {}


def _one(n):
    prm = orc.make_params(VARIANT, **kw)
    r = orc.solve_one(prm, G["l"], int(n), G["xp"][n], G["xv"][n], G["xa"][n], G["pf"][n])
    return n, r["status"], (int(r["info"][0]), int(r["info"][7]), int(r["info"][2])), (r["p"], r["v"], r["a"]) if r["status"] & 1 else None


if __name__ == "__main__":
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
    d = mp.Dmpc(VARIANT, **kw)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    rng = np.random.default_rng(17)
    t0 = time.time(); total = 0; bad = 0; worst = 0.0
    for step in range(2, 11):
        out = d.step_batch(l, xp, xv, xa, pf)
        inf, st = out["info"][0], out["status"][0]
        idx = np.unique(np.concatenate([np.argsort(inf[:, 4])[-per_step // 4:], np.where(inf[:, 2] > 1)[0], rng.integers(0, N, per_step)]))
        G.update(l=l[0], xp=xp[0], xv=xv[0], xa=xa[0], pf=pf[0])
        with get_context("fork").Pool(workers) as pool:
            res = pool.map(_one, idx.tolist(), chunksize=8)
        for n, rs, rec, traj in res:
            total += 1
            same = rs == st[n] and rec == (int(inf[n, 0]), int(inf[n, 1]), int(inf[n, 2]))
            e = 0.0
            if same and traj is not None:
                e = max(np.abs(traj[0] - out["p"][0, n]).max(), np.abs(traj[1] - out["v"][0, n]).max(), np.abs(traj[2] - out["a"][0, n]).max())
                worst = max(worst, e)
            if not same or e > TOL:
                bad += 1
                print(f"MISMATCH step {step} agent {n}: status {rs} vs {st[n]}, record {rec} vs {tuple(int(x) for x in inf[n, :3])}, l_inf {e:.2e}")
        print(f"step {step}: {idx.size} agents compared ({(inf[idx, 2] > 1).sum()} with ladder retries, most iterations {inf[idx, 4].max()}), {time.time() - t0:.0f} s", flush=True)
        ok = (out["status"] == 1)[..., None]
        l = np.where(ok, out["p"], l); xp = np.where(ok, out["p"][..., :3], xp)
        xv = np.where(ok, out["v"][..., :3], xv); xa = np.where(ok, out["a"][..., :3], xa)
    print(f"{total} agent-steps of the {CNAME} closed loop ({N} agents, {VARIANT}) compared with the oracle, {bad} mismatches, worst l_inf {worst:.2e}")
