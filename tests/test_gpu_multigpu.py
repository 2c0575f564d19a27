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


# ---- a box with two or more GPUs: the transports themselves, FIRST in this file (everything below needs one GPU only) --------------------
def test_two_real_ranks_over_rccl():
    """The wire itself: two processes, two GPUs, the library's RCCL all-gather between them (tests/rccl_two_ranks.py under
    torch.distributed.run): sharded transition (fp64 and mixed: the fp32 exchange), unequal clusters, the history gather and the post-checks
    on every rank, against the single-GPU run bit for bit.  Skipped on a box with one GPU (the emulated-rank tests above cover everything
    but the transport there)."""
    import os, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "tests", "rccl_two_ranks.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "RCCL_TWO_RANKS_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_two_real_gpus_in_one_process():
    """The other transport: ONE process, every visible GPU (dmpc_create(.., DMPC_DEVICE_ALL, ..): a host thread + stream + sub-context per GPU,
    hipMemcpyPeerAsync of every rank's chunk into every GPU's next table, events between the MPC steps) on REAL devices -- whole transitions
    (fp64 and mixed, unequal clusters), a second call on the same context, the post-checks on the histories gathered on the first GPU and
    dmpc_step_batch, against the single-GPU context bit for bit; also the bench's own closed loop through `python bench.py --gpus 2`.  Skipped
    on a box with one GPU (the emulated-rank tests below run the same protocol with the ranks sharing the device)."""
    import json, os, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    for precision, N in (("f64", 21), ("mixed", 20)):
        cfg, kw, po, pf = _scenes(N, 6, wl.SEED0 + 72)
        ref_d = mp.Dmpc("bound", precision=precision, **kw)
        ref = ref_d.transition(po, pf, 90, cfg["error_tol"])
        d = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, precision=precision, **kw)
        assert d.n_devices == torch.cuda.device_count()
        for rep in range(2):
            out = d.transition(po, pf, 90, cfg["error_tol"])
            assert np.array_equal(out["K_T_used"], ref["K_T_used"]) and np.array_equal(out["scene_status"], ref["scene_status"])
            for k in ("pk", "vk", "ak"):
                assert np.array_equal(out[k], ref[k]), (precision, rep, k)
        ok = ((ref["scene_status"] & mp.ST_REACHED) != 0).astype(np.int32)
        pc, pc_ref = d.postcheck(out["K_T_used"], pf, KT_alloc=90, mask=ok), ref_d.postcheck(ref["K_T_used"], pf, KT_alloc=90, mask=ok)
        for k in pc:
            assert np.array_equal(pc[k], pc_ref[k], equal_nan=True), k
        l, _, _ = ref_d.init_batch(po, pf); z = np.zeros_like(po)
        a, b = d.step_batch(l, po, z, z, pf), ref_d.step_batch(l, po, z, z, pf)
        for k in ("p", "v", "a", "status"):
            assert np.array_equal(a[k], b[k]), k
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DMPC_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["exchange_verified"] is True and line["config"]["n_ranks_seen"] == 2 and line["scaling"] == "strong"


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


# ---- one process, several GPUs (dmpc_create(prm, DMPC_DEVICE_ALL, ..)); the ranks emulated on this box's single GPU ----------------

@pytest.fixture
def emulated():
    yield mp.Dmpc.emulate_devices
    mp.Dmpc.emulate_devices(0)


@pytest.mark.parametrize("G,variant,N,precision", [(2, "bound", 20, "f64"), (3, "bound", 20, "f64"), (4, "bound2", 37, "f64"), (8, "bound", 100, "f64"),
                                                   (3, "bound", 20, "mixed")])
def test_group_context_transition_equals_single_gpu(emulated, G, variant, N, precision):
    """DMPC_DEVICE_ALL: threads + peer copies + events between the MPC steps; the same histories, step counts and verdicts as the
    single-GPU transition, bit for bit (f64; unequal clusters for G = 3, 8 | 20, 100; mixed precision: against the mixed single-GPU run --
    the exchanged table is fp32 there) -- and the post-checks run on the histories the group left resident on its first GPU"""
    cfg, kw, po, pf = _scenes(N, 4, wl.SEED0 + 41)
    ref_d = mp.Dmpc(variant, precision=precision, **kw)
    ref = ref_d.transition(po, pf, 100, cfg["error_tol"])
    emulated(G)
    d = mp.Dmpc(variant, device=mp.Dmpc.DEVICE_ALL, precision=precision, **kw)
    assert d.n_devices == G
    out = d.transition(po, pf, 100, cfg["error_tol"])
    assert np.array_equal(out["K_T_used"], ref["K_T_used"]) and np.array_equal(out["scene_status"], ref["scene_status"])
    for k in ("pk", "vk", "ak"):
        assert np.array_equal(out[k], ref[k]), k
    assert (ref["scene_status"] & mp.ST_REACHED).any()
    ok = (ref["scene_status"] & mp.ST_REACHED) != 0
    pc = d.postcheck(out["K_T_used"], pf, KT_alloc=100, mask=ok.astype(np.int32))          # resident, gathered on rank 0
    pc_ref = ref_d.postcheck(ref["K_T_used"], pf, KT_alloc=100, mask=ok.astype(np.int32))
    for k in pc:
        assert np.array_equal(pc[k], pc_ref[k], equal_nan=True), k
    # a second call on the same context (buffers, events and barrier state are reused)
    out2 = d.transition(po, pf, 100, cfg["error_tol"], histories=False)
    assert np.array_equal(out2["K_T_used"], ref["K_T_used"])


@pytest.mark.parametrize("G,N", [(8, 4), (8, 7), (4, 6)])
def test_group_context_with_fewer_agents_than_gpus(emulated, G, N):
    """the reference's small swarms (N = 4 .. 7) on a multi-GPU node: a DMPC_DEVICE_ALL context -- what the MEX gateway creates -- must
    run them (on its first GPU) instead of refusing to shard 4 agents over 8 GPUs; transition and single step alike"""
    cfg, kw, po, pf = _scenes(N, 3, wl.SEED0 + 45)
    ref = mp.Dmpc("bound", **kw).transition(po, pf, 100, cfg["error_tol"])
    emulated(G)
    d = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kw)
    assert d.n_devices == G
    out = d.transition(po, pf, 100, cfg["error_tol"])
    for k in ("pk", "vk", "ak", "K_T_used", "scene_status"):
        assert np.array_equal(out[k], ref[k]), k
    l, _, _ = d.init_batch(po, pf)
    z = np.zeros_like(po)
    assert np.array_equal(d.step_batch(l, po, z, z, pf)["p"], mp.Dmpc("bound", **kw).step_batch(l, po, z, z, pf)["p"])


@pytest.mark.parametrize("G,variant,N", [(2, "hard", 100), (3, "softall", 37), (5, "bound", 301)])
def test_group_context_step_batch_equals_single_gpu(emulated, G, variant, N):
    cfg, kw, po, pf = _scenes(N, 3, wl.SEED0 + 43, "C2" if variant == "hard" else "C4")
    d1 = mp.Dmpc(variant, **kw)
    l, _, _ = d1.init_batch(po, pf)
    z = np.zeros_like(po)
    one = d1.step_batch(l, po, z, z, pf)
    emulated(G)
    d = mp.Dmpc(variant, device=mp.Dmpc.DEVICE_ALL, **kw)
    out = d.step_batch(l, po, z, z, pf)
    for k in ("status", "p", "v", "a"):
        assert np.array_equal(out[k], one[k]), k
    assert np.array_equal(out["info"][..., :4], one["info"][..., :4])
    # the helper entry points run on the group's first GPU
    l2, _, _ = d.init_batch(po, pf)
    assert np.array_equal(l2, l)
    r = d.solve_one(l[0], 3, po[0, 3], z[0, 3], z[0, 3], pf[0, 3])
    assert r["status"] == one["status"][0, 3] and np.array_equal(r["p"], one["p"][0, 3])


def test_device_pointer_entry_points_refuse_a_group_context(emulated):
    import torch
    cfg, kw, po, pf = _scenes(10, 1, wl.SEED0 + 44)
    emulated(2)
    d = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kw)
    t = torch.zeros(8, dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.DmpcError, match="ONE GPU"):
        d.step_device(1, 1, 10, 0, *([t.data_ptr()] * 11))
    with pytest.raises(_lib.DmpcError):
        d.comm_init(b"\0" * 128, 2, 0)


def test_sharded_transition_gather_feeds_the_postcheck():
    """dmpc_transition_sharded_gather with a world of one RCCL rank: the all-gather of the padded history slabs and the re-packing
    run for real; dmpc_postcheck then reads the assembled scene-wide histories (f-1 after a sharded transition)"""
    cfg, kw, po, pf = _scenes(12, 3, wl.SEED0 + 45)
    ref_d = mp.Dmpc("bound", **kw)
    ref = ref_d.transition(po, pf, 80, cfg["error_tol"], histories=False)
    pc_ref = ref_d.postcheck(ref["K_T_used"], pf, KT_alloc=80)
    d = mp.Dmpc("bound", **kw)
    d.comm_init(mp.Dmpc.comm_unique_id(), 1, 0)
    out = d.transition_sharded(po, pf, 80, cfg["error_tol"], histories=False, gather=True)
    assert np.array_equal(out["K_T_used"], ref["K_T_used"])
    pc = d.postcheck(out["K_T_used"], pf, KT_alloc=80)
    for k in pc:
        assert np.array_equal(pc[k], pc_ref[k], equal_nan=True), k
    d.comm_destroy()


def test_group_context_large_batch_runs_as_two_groups(emulated):
    """64 scenes or more on a DMPC_DEVICE_ALL context: two groups of rank contexts side by side (the exchange of one half under the
    solve of the other); histories, verdicts and the post-checks on the resident histories equal the single-GPU run"""
    cfg, kw, po, pf = _scenes(16, 70, wl.SEED0 + 47)
    ref_d = mp.Dmpc("bound", **kw)
    ref = ref_d.transition(po, pf, 90, cfg["error_tol"])
    emulated(3)
    d = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kw)
    out = d.transition(po, pf, 90, cfg["error_tol"])
    for k in ("pk", "vk", "ak", "K_T_used", "scene_status"):
        assert np.array_equal(out[k], ref[k]), k
    ok = ((ref["scene_status"] & mp.ST_REACHED) != 0).astype(np.int32)
    pc, pc_ref = d.postcheck(out["K_T_used"], pf, KT_alloc=90, mask=ok), ref_d.postcheck(ref["K_T_used"], pf, KT_alloc=90, mask=ok)
    for k in pc:
        assert np.array_equal(pc[k], pc_ref[k], equal_nan=True), k
    one = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kw).debug_option("no_split", 1).transition(po, pf, 90, cfg["error_tol"], histories=False)
    assert np.array_equal(one["K_T_used"], ref["K_T_used"])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts the two ranks itself.  On this one-GPU box they share the
    GPU and the exchange is staged through gloo (a functional check of the multi-rank path, flagged as such in the JSON line): the headline
    scene (here 2 000 agents of it) sharded over the two ranks in a device-resident closed loop; the sharded step must equal the unsharded
    step of the same scene per agent, bit for bit, and every rank must hold the same gathered table (config.exchange_verified)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMPC_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--agents", "2000", "--no-secondary",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["exchange_verified"] is True and line["scaling"] == "strong"
    assert line["config"]["agents_per_scene"] == 2000 and line["config"]["agents_per_gpu"] == 1000 and "FUNCTIONAL CHECK" in line["data"]
    assert line["config"]["workload"].startswith("C4") and "setup_s" in line["config"] and "exchange_advance_and_gaps" in line["step_breakdown_ms"]
