"""CPU: workload generators (randomTest.m / randomExchange.m restatements) and host helpers."""
import numpy as np

from multiagent_planning_amd import workload as wl
from multiagent_planning_amd import api


def test_random_test_separation_and_bounds():
    cfg = wl.CONFIGS["C2"]
    kw = wl.solver_kwargs(cfg, 60)
    po, pf = wl.random_test(60, kw["pmin"], kw["pmax"], cfg["rmin_init"], cfg["c"], np.random.default_rng(5))
    e1 = np.array([1, 1, 1 / cfg["c"]])
    for pts in (po, pf):
        assert np.all(pts >= np.array(kw["pmin"])) and np.all(pts <= np.array(kw["pmax"]))
        d = np.sqrt((((pts[:, None] - pts[None]) * e1) ** 2).sum(-1)) + np.eye(60) * 10
        assert d.min() > cfg["rmin_init"]   # randomTest.m:18


def test_random_exchange_is_fixed_point_free_permutation():
    po, pf = wl.random_exchange(25, (-2.5, -2.5, 0.2), (2.5, 2.5, 2.2), 0.75, np.random.default_rng(7))
    idx = [int(np.where((po == g).all(axis=1))[0][0]) for g in pf]
    assert sorted(idx) == list(range(25))
    assert all(i != j for i, j in enumerate(idx))   # randomExchange.m:30-48


def test_density_box_matches_reference():
    pmin, pmax = wl.density_box(200)   # test/failure_rate.m:63-64, recorded in the golden workspace
    assert np.allclose(pmax, [2.92401774, 2.92401774, 6.04803548])
    assert np.allclose(pmin, [-2.92401774, -2.92401774, 0.2])


def test_c1_is_the_fixed_diagonal_swap():
    po, pf = wl.make_scenes(wl.CONFIGS["C1"], 1)
    assert po.shape == (1, 4, 3) and tuple(po[0, 0]) == (1.501, 1.5, 1.5) and tuple(pf[0, 0]) == (-1.5, -1.5, 1.5)


def test_small_helpers_fail_loudly_without_a_gpu():
    """the standalone helpers (initDMPC, is_inbounds, propStatedmpc, ReachedGoal, ...) run on the device like the rest
    of the path: on a box without a GPU they raise, they never fall back to host arithmetic."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_api.py")
    with pytest.raises(RuntimeError):
        api.is_inbounds([0, 0, 1], [-2.5, -2.5, 0.2], [2.5, 2.5, 2.2])
    with pytest.raises(RuntimeError):
        api.initDMPC([0, 0, 1], [10, 0, 1], 0.2, 15, 101)


def test_generator_restatement_properties():
    """oracle/generators.py (the checker of the device generators): separation, box, derangement, determinism."""
    from oracle import generators as G
    pmin, pmax = (-2.0, -2.0, 0.2), (2.0, 2.0, 2.2)
    po, pf = G.random_test(2, 25, pmin, pmax, 0.35, 2.0, 5)
    e1 = np.array([1, 1, 0.5])
    for p in (po[0], pf[1]):
        d = np.sqrt((((p[:, None] - p[None]) * e1) ** 2).sum(-1)) + np.eye(25) * 9
        assert d.min() > 0.35 and (p >= pmin).all() and (p <= pmax).all()
    again = G.random_test(2, 25, pmin, pmax, 0.35, 2.0, 5)
    assert np.array_equal(again[0], po) and np.array_equal(again[1], pf)
    xo, xf = G.random_exchange(3, 9, pmin, pmax, 0.5, 2)
    for s in range(3):
        idx = [int(np.where((xo[s] == g).all(axis=1))[0][0]) for g in xf[s]]
        assert sorted(idx) == list(range(9)) and all(i != j for i, j in enumerate(idx))
