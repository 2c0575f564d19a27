"""Synthetic workloads: the reference's start/goal generators and experiment constants.

  random_test      -- dmpc/matlab/randomTest.m:1-56   (rejection sampling, min ellipsoidal separation)
  random_exchange  -- dmpc/matlab/randomExchange.m:1-55 (starts + a fixed-point-free permutation as goals)
  CONFIGS          -- the constants blocks of the reference scripts behind BASELINE.json's configs
                      

The reference draws from MATLAB's global `rand` stream; here a seeded numpy Generator is used
(seed 20180926 + config index by convention), so workloads are reproducible but not the
reference's own random numbers.
"""
import numpy as np

SEED0 = 20180926


def density_box(N):
    """Workspace that keeps the agent density constant (test/failure_rate.m:63-64)."""
    s = float(N) ** (1.0 / 3.0)
    return (-s / 2, -s / 2, 0.2), (s / 2, s / 2, s + 0.2)


def _sample_separated(rng, N, pmin, pmax, rmin, e1, max_iter=200000):
    pmin, pmax = np.asarray(pmin, float), np.asarray(pmax, float)
    while True:   # randomTest.m:7-28: restart from scratch if a point cannot be placed
        pts = np.empty((N, 3))
        pts[0] = pmin + (pmax - pmin) * rng.random(3)
        ok = True
        for n in range(1, N):
            tries = 0
            while True:
                cand = pmin + (pmax - pmin) * rng.random(3)
                d = np.sqrt((((pts[:n] - cand) * e1) ** 2).sum(axis=1))
                tries += 1
                if (d > rmin).all():
                    pts[n] = cand
                    break
                if tries > max_iter:
                    ok = False
                    break
            if not ok:
                break
        if ok:
            return pts


def random_test(N, pmin, pmax, rmin, c=1.0, rng=None):
    """[po,pf] = randomTest(N,pmin,pmax,rmin,E1,order) with E1 = diag(1,1,1/c), order = 2."""
    rng = rng or np.random.default_rng(SEED0)
    e1 = np.array([1.0, 1.0, 1.0 / c])
    po = _sample_separated(rng, N, pmin, pmax, rmin, e1)
    pf = _sample_separated(rng, N, pmin, pmax, rmin, e1)
    return po, pf


def random_exchange(N, pmin, pmax, rmin, rng=None):
    """[po,pf] = randomExchange(N,pmin,pmax,rmin): Euclidean separation; goals = starts permuted
    so that no agent keeps its own start (randomExchange.m:30-52)."""
    rng = rng or np.random.default_rng(SEED0)
    po = _sample_separated(rng, N, pmin, pmax, rmin, np.ones(3))
    array = list(range(N))
    perm = [0] * N
    for i in range(N):
        aux = [x for x in array if x != i]
        if i == N - 1:
            perm[i] = array[0]
        elif i == N - 2 and aux[-1] == N - 1:
            perm[i] = N - 1
            array.remove(N - 1)
        else:
            j = int(rng.integers(0, N - 1 - i))      # randi([1 N-i]) with the .m's 1-based i
            perm[i] = aux[j]
            array.remove(aux[j])
    return po, po[perm]


# constants of the reference scripts (file:line in the comments)
CONFIGS = {
    # dmpc/matlab/dmpc_soft_bound.m:7-78 -- fixed 4-agent diagonal swap
    "C1": dict(variant="bound", N=4, h=0.2, rmin=0.5, c=1.5, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4,
               pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2), error_tol=0.05, K_T=101,
               po=[(1.501, 1.5, 1.5), (-1.5, -1.5, 1.5), (-1.5, 1.5, 1.5), (1.5, -1.5, 1.5)],
               pf=[(-1.5, -1.5, 1.5), (1.5, 1.5, 1.5), (1.5, -1.5, 1.5), (-1.5, 1.5, 1.5)]),
    # test/comp_hardsoft2.m:7-97 -- hard ellipsoidal constraints, density-scaled box, randomTest
    "C2": dict(variant="hard", N=100, h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4,
               rmin_init=0.35, error_tol=0.01, K_T=101, generator="random_test", box="density"),
    # test/success_test_softdmpc.m:7-74 -- all-neighbour slack, randomExchange (box scaled at constant density)
    "C3": dict(variant="softall", N=1000, h=0.2, rmin=0.5, c=1.5, alim=0.5, Q1=1000.0, S1=100.0, term=-1e5,
               rmin_init=0.75, error_tol=0.05, K_T=101, generator="random_exchange", box="density"),
    # test/failure_rate.m:7-99 -- the primary soft variant, sharded across GPUs
    "C4": dict(variant="bound", N=10000, h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4,
               rmin_init=0.35, error_tol=0.01, K_T=150, generator="random_test", box="density"),   # max_K = 151, `k < max_K`: 150 columns
    # test/comp_repair.m:19-29,93 -- repair heuristic, dense box
    "C5": dict(variant="repair", N=200, h=0.2, rmin=0.5, c=1.5, alim=0.5, Q1=1000.0, S1=100.0, term=-1e6,
               rmin_init=0.75, error_tol=0.05, K_T=101, generator="random_test", box="density"),
}


def solver_kwargs(cfg, N=None):
    """keyword arguments for multiagent_planning_amd.Dmpc / oracle.make_params from a CONFIGS entry."""
    N = N or cfg["N"]
    if cfg.get("box") == "density":
        pmin, pmax = density_box(N)
    else:
        pmin, pmax = cfg["pmin"], cfg["pmax"]
    return dict(h=cfg["h"], rmin=cfg["rmin"], c=cfg["c"], alim=cfg["alim"], Q1=cfg["Q1"], S1=cfg["S1"],
                term=cfg["term"], pmin=tuple(pmin), pmax=tuple(pmax))


def make_scenes(cfg, S, N=None, seed=None):
    """S independent start/goal sets of a config: (po, pf) each [S,N,3]."""
    N = N or cfg["N"]
    kw = solver_kwargs(cfg, N)
    rng = np.random.default_rng(SEED0 if seed is None else seed)
    po, pf = np.empty((S, N, 3)), np.empty((S, N, 3))
    for s in range(S):
        if "po" in cfg:
            po[s], pf[s] = np.asarray(cfg["po"], float), np.asarray(cfg["pf"], float)
        elif cfg["generator"] == "random_exchange":
            po[s], pf[s] = random_exchange(N, kw["pmin"], kw["pmax"], cfg["rmin_init"], rng)
        else:
            po[s], pf[s] = random_test(N, kw["pmin"], kw["pmax"], cfg["rmin_init"], cfg["c"], rng)
    return po, pf


def make_scenes_device(dmpc, cfg, S, N=None, seed=None):
    """make_scenes with the rejection sampling done on the GPU (dmpc_random_test / dmpc_random_exchange): S scenes in
    one launch, one wave per point set.  Same algorithm, its own counter-based stream (seeded like make_scenes)."""
    N = N or cfg["N"]
    kw = solver_kwargs(cfg, N)
    seed = SEED0 if seed is None else seed
    if cfg.get("generator") == "random_exchange":
        return dmpc.random_exchange(S, N, kw["pmin"], kw["pmax"], cfg["rmin_init"], seed)
    return dmpc.random_test(S, N, kw["pmin"], kw["pmax"], cfg["rmin_init"], cfg["c"], seed)
