"""GPU (one device): the in-library multi-GPU path (dmpc_multigpu.hip).  A box with one GPU can check everything but the
wire: the sharded transition with one rank -- with and without an RCCL communicator (ncclAllGather with nranks = 1 runs the
real collective calls) -- against dmpc_transition bit for bit, and the sharded STEP with the ranks of a 2-, 3- and 8-rank
job run one after the other on this GPU (dmpc_debug_set_rank), unequal clusters included, against the unsharded step."""
import ctypes as C

import numpy as np
import pytest

import multiagent_planning_amd as mp
from multiagent_planning_amd import _lib, workload as wl

pytestmark = pytest.mark.gpu


def _scenes(N, S, seed, cfgname="C4"):
    cfg = wl.CONFIGS[cfgname]
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, S, N, seed)
    return cfg, kw, po, pf


@pytest.mark.parametrize("with_comm", [False, True])
def test_sharded_transition_one_rank_equals_dmpc_transition(with_comm):
    cfg, kw, po, pf = _scenes(20, 3, wl.SEED0 + 21)
    ref = mp.Dmpc("bound", **kw).transition(po, pf, 60, cfg["error_tol"])
    d = mp.Dmpc("bound", **kw)
    if with_comm:
        d.comm_init(mp.Dmpc.comm_unique_id(), 1, 0)      # a real RCCL communicator of one rank
    out = d.transition_sharded(po, pf, 60, cfg["error_tol"])
    assert np.array_equal(out["K_T_used"], ref["K_T_used"]) and np.array_equal(out["scene_status"], ref["scene_status"])
    for k in ("pk", "vk", "ak"):
        assert np.array_equal(out[k], ref[k]), k
    assert (ref["scene_status"] & mp.ST_REACHED).any() and (ref["K_T_used"] < 60).any()
    if with_comm:
        d.comm_destroy()


@pytest.mark.parametrize("variant,N,G", [("bound", 10, 3), ("hard", 100, 8), ("softall", 37, 2), ("hard", 301, 4), ("bound", 700, 3), ("ondemand", 515, 8)])
def test_sharded_step_ranks_in_turn_equal_the_unsharded_step(variant, N, G):
    """every rank of a G-rank job solves its cluster (dmpc.cpp:1600-1625: the first N mod G clusters one agent more) against the
    padded rank-major table; the union of the ranks' outputs and the exchanged next table equal the one-rank step bit for bit"""
    import torch
    cfg, kw, po, pf = _scenes(N, 2, wl.SEED0 + 31, "C2" if variant == "hard" else "C4")
    S = 2
    d1 = mp.Dmpc(variant, **kw)
    l, _, _ = d1.init_batch(po, pf)
    z = np.zeros_like(po)
    one = d1.step_batch(l, po, z, z, pf)
    dev = torch.device("cuda", 0)
    parts = [_lib.partition(N, G, r) for r in range(G)]
    cmax = parts[0][2]
    lT = np.zeros((G, S, 45, cmax))
    for r, (lo, cnt, _) in enumerate(parts):
        lT[r, :, :, :cnt] = l[:, lo:lo + cnt].transpose(0, 2, 1)
    lT_d = torch.from_numpy(lT).to(dev)
    lT_next = torch.zeros_like(lT_d)
    L = _lib.load()
    L.dmpc_debug_set_rank.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for r, (lo, cnt, _) in enumerate(parts):
        d = mp.Dmpc(variant, **kw)
        assert L.dmpc_debug_set_rank(d._ctx, G, r) == 0
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, lo:lo + cnt])).to(dev)
        xp, xv, xa, gf = t(po), t(z), t(z), t(pf)
        p = torch.empty((S, cnt, 45), dtype=torch.float64, device=dev); v, a = torch.empty_like(p), torch.empty_like(p)
        st = torch.zeros((S, cnt), dtype=torch.int32, device=dev); inf = torch.zeros((S, cnt, 8), dtype=torch.int32, device=dev)
        d.step_sharded_device(S, N, lT_d.data_ptr(), xp.data_ptr(), xv.data_ptr(), xa.data_ptr(), gf.data_ptr(), p.data_ptr(), v.data_ptr(),
                              a.data_ptr(), lT_next.data_ptr(), st.data_ptr(), inf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sl = slice(lo, lo + cnt)
        assert np.array_equal(st.cpu().numpy(), one["status"][:, sl]), (variant, r)
        assert np.array_equal(p.cpu().numpy(), one["p"][:, sl]) and np.array_equal(a.cpu().numpy(), one["a"][:, sl]), (variant, r)
        assert np.array_equal(inf.cpu().numpy()[..., :4], one["info"][:, sl, :4]), (variant, r)
        nxt = lT_next[r].cpu().numpy()[:, :, :cnt].transpose(0, 2, 1)          # this rank's slot of the exchanged table
        ok = one["status"][:, sl] & 1 == 1
        want = np.where(ok[..., None], one["p"][:, sl], l[:, sl])              # unsolved agents keep their old prediction
        assert np.array_equal(nxt, want), (variant, r)
