"""development campaign (not collected): the cell-grid neighbour lists against the whole-table walk on random mid-size scenes -- random agent
counts, densities (the workspace scaled), variants with a finite neighbour radius, fp64 and mixed precision, closed loops of several MPC steps
(fast agents late in a loop are what stresses the chord pre-test: their segments are long and curved).  Every output must be identical, bit
for bit.   usage: python tests/dev/gpu_grid_campaign.py [scenes] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl

nscen = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
VARS = ["bound", "bound2", "all3", "ondemand", "hard", "cpp", "cpp2"]


def steps(variant, kw, po, pf, nsteps, precision, **opts):
    d = mp.Dmpc(variant, precision=precision, **kw)
    for k_, v_ in opts.items():
        d.debug_option(k_, v_)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    outs = []
    for _ in range(nsteps):
        o = d.step_batch(l, xp, xv, xa, pf)
        outs.append(o)
        ok = (o["status"] == 1)[..., None]
        l = np.where(ok, o["p"], l); xp = np.where(ok, o["p"][..., :3], xp)
        xv = np.where(ok, o["v"][..., :3], xv); xa = np.where(ok, o["a"][..., :3], xa)
    return outs


t0 = time.time(); bad = 0; agent_steps = 0
for it in range(nscen):
    variant = VARS[rng.integers(len(VARS))]
    precision = "mixed" if rng.random() < 0.25 else "f64"
    N = int(rng.integers(200, 1600)); S = int(rng.integers(1, 3))
    cfg = dict(wl.CONFIGS["C4"])
    cfg["alim"] = float(rng.choice([0.5, 1.0, 2.0]))
    kw = wl.solver_kwargs(cfg, N)
    dens = float(rng.choice([0.6, 1.0, 1.6]))          # workspace side scaled: denser / sparser than the reference's constant density
    pmin, pmax = np.array(kw["pmin"]), np.array(kw["pmax"])
    ctr = 0.5 * (pmin + pmax); kw["pmin"] = tuple(ctr + (pmin - ctr) * dens); kw["pmax"] = tuple(ctr + (pmax - ctr) * dens)
    po = np.array(kw["pmin"]) + rng.random((S, N, 3)) * (np.array(kw["pmax"]) - np.array(kw["pmin"]))
    pf = np.array(kw["pmin"]) + rng.random((S, N, 3)) * (np.array(kw["pmax"]) - np.array(kw["pmin"]))
    nst = int(rng.integers(2, 7))
    g = steps(variant, kw, po, pf, nst, precision, grid_min=64)
    b = steps(variant, kw, po, pf, nst, precision, no_cull=1)
    for k, (x, y) in enumerate(zip(g, b)):
        agent_steps += S * N
        for key in ("status", "info", "p", "v", "a"):
            if not np.array_equal(x[key], y[key]):
                bad += 1
                print(f"MISMATCH scene {it} {variant} {precision} N={N} S={S} dens {dens} alim {cfg['alim']} step {k + 2}: {key}")
                break
print(f"{nscen} scenes, {agent_steps} agent-steps compared, {bad} mismatching steps, {time.time() - t0:.0f} s")
