"""development (not collected): ONE (seed, scene, variant) of tests/dev/gpu_campaign.py again -- the random stream is replayed up to that scene, the
variant's teacher-forced steps are run on the GPU and the oracle, and at every step the agents that differ are listed with their branch records and,
for the worst one, the objective value and the KKT certificate (tests/certificates.py) of BOTH answers: which of the two holds the minimiser.
usage: python tests/dev/gpu_campaign_scene.py SEED SCENE VARIANT        (run under tools/with_lib.py to look at another build of the library)"""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
from oracle import oracle as orc
from helpers import ALL_VARIANTS, init_table
import certificates as cert

seed0, scene, want = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed0)
for it in range(scene + 1):
    N = int(rng.integers(2, 90))
    dense = rng.random() < 0.5
    cfgname = "C5" if dense else "C2"
    cfg = wl.CONFIGS[cfgname]
    kw = wl.solver_kwargs(cfg, N)
    if rng.random() < 0.3:
        s = 0.8
        kw["pmin"] = tuple(np.asarray(kw["pmin"]) * [s, s, 1]); kw["pmax"] = tuple(np.asarray(kw["pmax"]) * [s, s, 1])
    sc_seed = int(rng.integers(1 << 30))
    nsteps = {v: int(rng.integers(2, 7)) for v in ALL_VARIANTS}
po, pf = wl.make_scenes(dict(cfg), 1, N, sc_seed); po, pf = po[0], pf[0]
print(f"seed {seed0} scene {scene}: N = {N}, box {cfgname}, variant {want}, {nsteps[want]} steps")
d = mp.Dmpc(want, **kw); prm = orc.make_params(want, **kw)
for kv in filter(None, os.environ.get("DBG_OPTS", "").split(",")):   # development options of the context, e.g. DBG_OPTS=reduced_solver=0
    d.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
l = init_table(po, pf); xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(nsteps[want]):
    out = d.step_batch(l, xp, xv, xa, pf); ref = orc.step(prm, l, xp, xv, xa, pf, nthreads=8)
    ok = (ref["status"] & 1) == 1
    e = np.zeros(N)
    for key in ("p", "v", "a"):
        e = np.maximum(e, np.abs(out[key] - ref[key]).max(axis=1) * ok)
    print(f"step {k + 2}: status equal {np.array_equal(out['status'], ref['status'])}, ladder counts equal {np.array_equal(out['info'][:, 2], ref['info'][:, 2])}, worst l_inf {e.max():.2e} (agent {int(e.argmax())})")
    for n in np.argsort(e)[::-1][:3]:
        if e[n] < 1e-9: break
        qp = orc.assemble_one(prm, l, int(n), xp[n], xv[n], xa[n], pf[n], level=int(ref["info"][n, 2]) - 1)
        obj = lambda a: float(0.5 * (x := cert.complete_slack(qp, a)) @ qp["H"] @ x + qp["f"] @ x)
        cg, co = cert.kkt_certificate(qp, out["a"][n]), cert.kkt_certificate(qp, ref["a"][n])
        print(f"   agent {n}: l_inf {e[n]:.2e}; GPU info {out['info'][n]} | oracle info {ref['info'][n]}")
        print(f"      objective GPU {obj(out['a'][n]):.10f}  oracle {obj(ref['a'][n]):.10f}   (lower is the minimiser)")
        print(f"      KKT GPU: primal {cg['primal']:.1e} stationarity {cg['stat_rel']:.1e} active {cg['n_active']} | oracle: primal {co['primal']:.1e} stationarity {co['stat_rel']:.1e} active {co['n_active']}")
    okb = out["status"] & 1 == 1
    l = np.where(okb[:, None], out["p"], l); xp = np.where(okb[:, None], out["p"][:, :3], xp)
    xv = np.where(okb[:, None], out["v"][:, :3], xv); xa = np.where(okb[:, None], out["a"][:, :3], xa)
