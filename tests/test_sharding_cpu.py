"""CPU: the N>1 path (agent sharding + per-step all-gather) with world_size-2 gloo processes.

The product's local solver is the HIP kernel; here the oracle is injected as the local solver so
that the partition / table layout / collective logic of multiagent_planning_amd.driver is exercised
without a GPU.  The sharded result must be bit-identical to the unsharded one (Jacobi update:
no order dependence between agents)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

from helpers import ROOT, load_golden, step14_inputs
from multiagent_planning_amd import driver


def test_partition_matches_reference_clusters():
    # dmpc.cpp:1600-1625: N/G each, first N mod G clusters get one more
    assert driver.partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert driver.partition(8, 2) == [(0, 4), (4, 8)]
    assert driver.partition(5, 8)[:6] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5)]


def test_chunked_layout_roundtrip():
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((3, 8, 45))
    lT = driver.rows_to_chunked(rows, 4)
    assert lT.shape == (4, 3, 45, 2)
    assert np.array_equal(driver.chunked_to_rows(lT), rows)
    assert lT[2, 1, 7, 1] == rows[1, 2 * 2 + 1, 7]


def _oracle_local_step(kw, variant, G, N=None):
    from oracle import oracle as orc
    prm = orc.make_params(variant, **kw)

    def local_step(lT_full, x_p, x_v, x_a, pf, g_local):
        rows = driver.chunked_to_rows(np.asarray(lT_full), N)
        S, Nn, _ = rows.shape
        lo, hi = driver.partition(Nn, G)[g_local]
        C = hi - lo
        p = np.zeros((S, C, 45)); v = np.zeros_like(p); a = np.zeros_like(p)
        status = np.zeros((S, C), dtype=np.int32)
        for s in range(S):
            for ci in range(C):
                n = lo + ci
                r = orc.solve_one(prm, rows[s], n, x_p[s, ci], x_v[s, ci], x_a[s, ci], pf[s, ci])
                status[s, ci] = r["status"]
                if r["status"] & 1:
                    p[s, ci], v[s, ci], a[s, ci] = r["p"], r["v"], r["a"]
        return dict(p=p, v=v, a=a, status=status)
    return local_step


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g)
    N = 16   # small scene cut from the recorded one (agents 0..15 see only each other)
    rows = l[None, :N]
    lT = driver.rows_to_chunked(rows, world)
    C = N // world
    sl = slice(rank * C, (rank + 1) * C)
    stepper = driver.ShardedStepper(_oracle_local_step(kw, "bound2", world), rank, world)
    x_p, x_v, x_a, p_f = xp[None, sl], xv[None, sl], xa[None, sl], pf[None, sl]
    for it in range(2):   # two consecutive MPC steps: the gathered table feeds the next step
        lT, out, failed = stepper.step(lT, x_p, x_v, x_a, p_f)
        ok = (out["status"] & 1) == 1
        x_p = np.where(ok[..., None], out["p"][..., :3], x_p)
        x_v = np.where(ok[..., None], out["v"][..., :3], x_v)
        x_a = np.where(ok[..., None], out["a"][..., :3], x_a)
    np.save(os.path.join(out_dir, f"lT_{rank}.npy"), np.asarray(lT))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_rank(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    tmp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "lT_0.npy"), np.load(tmp_path / "lT_1.npy")
    assert np.array_equal(a, b)   # every rank holds the same full table after the all-gather
    # single-rank reference run of the same two steps
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g)
    N = 16
    lT = driver.rows_to_chunked(l[None, :N], 1)
    stepper = driver.ShardedStepper(_oracle_local_step(kw, "bound2", 1), 0, 1)
    x_p, x_v, x_a, p_f = xp[None, :N], xv[None, :N], xa[None, :N], pf[None, :N]
    for it in range(2):
        lT, out, failed = stepper.step(lT, x_p, x_v, x_a, p_f)
        ok = (out["status"] & 1) == 1
        x_p = np.where(ok[..., None], out["p"][..., :3], x_p)
        x_v = np.where(ok[..., None], out["v"][..., :3], x_v)
        x_a = np.where(ok[..., None], out["a"][..., :3], x_a)
    assert np.array_equal(driver.chunked_to_rows(a), driver.chunked_to_rows(np.asarray(lT)))


def _transition_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from multiagent_planning_amd import workload as wl
    from helpers import init_table
    cfg = wl.CONFIGS["C1"]
    kw = wl.solver_kwargs(cfg)
    po, pf = (np.asarray(cfg[k], float) for k in ("po", "pf"))
    N = 4
    C = N // world
    sl = slice(rank * C, (rank + 1) * C)
    lT = driver.rows_to_chunked(init_table(po, pf)[None], world)
    stepper = driver.ShardedStepper(_oracle_local_step(kw, "bound", world), rank, world)
    z = np.zeros((1, C, 3))
    res = driver.run_transition_sharded(stepper, lT, po[None, sl], z, z, pf[None, sl], 60, cfg["error_tol"])
    np.savez(os.path.join(out_dir, f"tr_{world}_{rank}.npz"), pk=res["pk"], used=res["K_T_used"], reached=res["reached"],
             failed=res["failed"], lT=np.asarray(res["lT"]))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_closed_loop_transition_equals_single_rank(tmp_path):
    """whole C1 transition (dmpc_soft_bound.m constants, 4 agents) with the agents split over 2 gloo ranks: same number of
    MPC steps, same termination decision, bit-identical histories and final table as the unsharded loop."""
    port = 31500 + (os.getpid() % 2000)
    tmp.spawn(_transition_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    _transition_worker(0, 1, port + 1, str(tmp_path))
    one = np.load(tmp_path / "tr_1_0.npz")
    r0, r1 = np.load(tmp_path / "tr_2_0.npz"), np.load(tmp_path / "tr_2_1.npz")
    assert one["reached"].all() and not one["failed"].any() and 20 < one["used"][0] < 60
    for r in (r0, r1):
        assert np.array_equal(r["used"], one["used"]) and np.array_equal(r["reached"], one["reached"]) and np.array_equal(r["failed"], one["failed"])
    assert np.array_equal(np.concatenate([r0["pk"], r1["pk"]], axis=1), one["pk"])
    assert np.array_equal(driver.chunked_to_rows(r0["lT"]), driver.chunked_to_rows(one["lT"])) and np.array_equal(r0["lT"], r1["lT"])


def _unequal_worker(rank, world, port, out_dir, shrink):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from multiagent_planning_amd import workload as wl
    from helpers import init_table
    cfg = wl.CONFIGS["C4"]
    N = 7
    kw = wl.solver_kwargs(cfg, 20)
    if shrink:   # a workspace the straight-line starts leave at once: SOLVED|OUTBOUND on the first solve
        kw["pmax"] = (kw["pmax"][0], kw["pmax"][1], 0.6 * kw["pmax"][2])
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 77)
    po, pf = po[0], pf[0]
    lo, hi = driver.partition(N, world)[rank]
    lT = driver.rows_to_chunked(init_table(po, pf)[None], world)
    stepper = driver.ShardedStepper(_oracle_local_step(kw, "bound2" if shrink else "bound", world, N), rank, world)
    z = np.zeros((1, hi - lo, 3))
    res = driver.run_transition_sharded(stepper, lT, po[None, lo:hi], z, z, pf[None, lo:hi], 12, cfg["error_tol"])
    np.savez(os.path.join(out_dir, f"un_{int(shrink)}_{world}_{rank}.npz"), pk=res["pk"], used=res["K_T_used"], reached=res["reached"],
             failed=res["failed"], rows=driver.chunked_to_rows(np.asarray(res["lT"]), N))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shrink", [False, True])
def test_three_rank_unequal_clusters_equal_single_rank(tmp_path, shrink):
    """7 agents on 3 gloo ranks -- clusters of 3, 2, 2 agents as dmpc.cpp:1600-1625, padded rank-major table -- against the
    unsharded loop: same histories, table and stop decision.  shrink: the workspace is cut so that agents leave it on their
    first solve; bound2 reports SOLVED|OUTBOUND and the trial must stop there on EVERY rank (the abort rule shared with
    dmpc_transition: any status other than exactly SOLVED)."""
    port = 33500 + (os.getpid() % 2000) + (7 if shrink else 0)
    tmp.spawn(_unequal_worker, args=(3, port, str(tmp_path), shrink), nprocs=3, join=True)
    _unequal_worker(0, 1, port + 1, str(tmp_path), shrink)
    one = np.load(tmp_path / f"un_{int(shrink)}_1_0.npz")
    rs = [np.load(tmp_path / f"un_{int(shrink)}_3_{r}.npz") for r in range(3)]
    for r in rs:
        assert np.array_equal(r["used"], one["used"]) and np.array_equal(r["failed"], one["failed"]) and np.array_equal(r["reached"], one["reached"])
        assert np.array_equal(r["rows"], one["rows"])
    assert np.array_equal(np.concatenate([r["pk"] for r in rs], axis=1), one["pk"])
    if shrink:
        assert one["failed"].all() and one["used"][0] == 2     # stopped at the first solve
    else:
        assert not one["failed"].any() and one["used"][0] == 12
