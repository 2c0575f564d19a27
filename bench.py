#!/usr/bin/env python3
"""bench.py -- agent-QP solves/sec of the DMPC per-agent horizon QP (K = 15) on MI355X.

Contract (one JSON line on rank 0):  python bench.py --gpus N --steps K --warmup W
  * a "step" = ONE MPC step of the hot path (scan + rows + QP + propagate for every agent) over a
    batch of S independent scenes ("trials", test/comp_hardsoft2.m:12; default S = 512) whose tables and states are
    already resident in HBM; steady-state replay of a captured congested MPC step
  * workload at 1 GPU = BASELINE.json configs[1]: 100 agents/scene, hard ellipsoidal constraints
    (solveHardDMPC, constants of test/comp_hardsoft2.m), S scenes batched
  * N > 1 GPUs: one process per GPU (torch.distributed / RCCL); every scene has 100*N agents in a
    density-scaled box, rank r owns agents [100r, 100r+100) of every scene (contiguous clusters as
    dmpc/cpp/dmpc.cpp:1600-1625) and each step ends with ONE all-gather of the new predictions
    (the `l = new_l` / `prev_obs = obs` exchange).  Per-GPU work is fixed -> "scaling": "weak".
  * value = solves of all ranks / max-over-ranks wall time between barriers.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def capture_state(dmpc, cfg, S, N, k_cap, seed):
    """Closed-loop run of S scenes up to MPC step k_cap on the GPU (host-array API); returns the
    table and states that are the INPUT of step k_cap+1.  Scenes that abort earlier (an agent
    infeasible / collided, the reference `break`s the trial) keep the last valid state."""
    from multiagent_planning_amd import workload as wl
    po, pf = wl.make_scenes(cfg, S, N, seed)
    l, _, _ = dmpc.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    alive = np.ones(S, bool)
    for k in range(1, k_cap):
        out = dmpc.step_batch(l, xp, xv, xa, pf)
        ok = (out["status"] == 1)
        alive &= ok.all(axis=1)
        upd = alive[:, None] & ok
        l = np.where(upd[..., None], out["p"], l)
        xp = np.where(upd[..., None], out["p"][..., :3], xp)
        xv = np.where(upd[..., None], out["v"][..., :3], xv)
        xa = np.where(upd[..., None], out["a"][..., :3], xa)
    return l, xp, xv, xa, pf, alive


PMC_SUMMARY = "r04_pmc_summary.json"


def source_hash():
    """sha256 over the kernel sources (the key a committed counter summary must carry to be quoted next to a live measurement)"""
    import hashlib
    d = os.path.join(ROOT, "multiagent_planning_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scenes", type=int, default=512,
                    help="independent scenes (trials) batched per step; 512 ~ one Monte-Carlo experiment of the reference "
                         "(test/comp_hardsoft2.m: 10 swarm sizes x 50 trials)")
    ap.add_argument("--agents-per-gpu", type=int, default=100)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--capture-step", type=int, default=12,
                    help="MPC step whose inputs are replayed; scenes that abort earlier keep their last valid state "
                         "(solveHardDMPC on C2 aborts every scene at its first solve, so the replayed state is step 2)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--emulate-gpus", type=int, default=0,
                    help="tuning aid: on ONE GPU run rank 0's share of a G-GPU job (table of 100*G agents per scene, "
                         "no collective) to see the per-rank step time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="development: a dmpc_debug_option of the headline context (A/B runs, e.g. grid_min=1073741824); recorded in the line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import multiagent_planning_amd as mp
    from multiagent_planning_amd import workload as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_gpus:
        # launched as `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` does (one process per GPU, RCCL).
        # On a box with fewer GPUs than N this is only a functional check of the multi-rank path (all ranks share GPU 0, the exchange
        # staged through gloo; the JSON line says so) and has to be asked for with DMPC_BENCH_SHARE_GPU=1.
        import socket
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not os.environ.get("DMPC_BENCH_SHARE_GPU"):
            sys.exit(f"bench.py --gpus {args.gpus}: {ndev} GPU(s) visible (DMPC_BENCH_SHARE_GPU=1 runs the ranks on one GPU as a functional check)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and not args.emulate_gpus:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    G = world
    emu = args.emulate_gpus if (world == 1 and args.emulate_gpus > 1) else 0
    if emu:
        G = emu
    # functional check of the multi-rank path on a box with ONE GPU (tests only, never a reported number): all ranks
    # share cuda:0 and the exchange is staged through gloo
    share = bool(os.environ.get("DMPC_BENCH_SHARE_GPU")) and world > 1
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # DMPC_BENCH_FORCE_DIST=1: run the RCCL calls of the multi-rank path with a world of one (functional check of
    # the collectives' arguments on a single-GPU box)
    use_dist = (G > 1 and not emu) or (bool(os.environ.get("DMPC_BENCH_FORCE_DIST")) and not emu)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=G)
        else:   # device_id binds the RCCL communicator to this rank's GPU up front (no device guessing in barrier())
            dist.init_process_group("nccl", rank=rank, world_size=G, device_id=dev)

    cfg = wl.CONFIGS[args.config]
    C = args.agents_per_gpu
    N = C * G
    S = args.scenes
    kw = wl.solver_kwargs(cfg, N)
    dmpc = mp.Dmpc(cfg["variant"], device=local_rank, **kw)
    for kv in args.debug_option:
        dmpc.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    # the per-step exchange runs INSIDE the library (dmpc_step_sharded_device: solve + ncclAllGather on one stream); the RCCL id
    # of the library's communicator travels through torch.distributed.  If the library cannot set its communicator up, the
    # exchange falls back to torch.distributed's all-gather (and the JSON line says so).
    exchange = "none"
    if use_dist and not share:
        exchange = "torch.distributed all_gather_into_tensor"
        # (every rank takes part in every collective below whatever fails locally: a rank that skipped one would hang the others)
        idt = torch.zeros(129, dtype=torch.uint8, device=dev)      # 128-byte id + "rank 0 has one" flag
        if rank == 0:
            try:
                idt[:128].copy_(torch.frombuffer(bytearray(mp.Dmpc.comm_unique_id()), dtype=torch.uint8))
                idt[128] = 1
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] rank 0: no RCCL id from the library ({e})\n")
        dist.broadcast(idt, src=0)
        okf = torch.zeros(1, dtype=torch.int32, device=dev)
        have_id = torch.tensor([int(idt[128].item())], dtype=torch.int32, device=dev)
        if int(have_id.item()) == 1:      # the same on every rank: all of them enter comm_init (a collective) or none
            try:
                dmpc.comm_init(bytes(idt[:128].cpu().numpy().tobytes()), G, rank)
                okf = torch.ones(1, dtype=torch.int32, device=dev)
                exchange = "in-library RCCL all-gather (dmpc_step_sharded_device)"
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] rank {rank}: library communicator unavailable ({e}); using torch.distributed\n")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)     # all ranks take the same path
        if int(okf.item()) == 0:
            exchange = "torch.distributed all_gather_into_tensor"
    elif share:
        exchange = "gloo (shared-GPU functional check)"
    in_lib = exchange.startswith("in-library")

    # ---- synthetic inputs: captured congested step (identical on every rank: deterministic) ----
    l, xp, xv, xa, pf, alive = capture_state(dmpc, cfg, S, N, args.capture_step, wl.SEED0 + 2)
    # device-resident buffers in the kernel's layouts
    def dev_t(a, dtype=torch.float64):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype)
    rows = dev_t(l)                                                     # [S][N][45]
    lT = torch.empty((G, S, 45, C), dtype=torch.float64, device=dev)    # chunked transposed table
    stream = torch.cuda.current_stream().cuda_stream
    dmpc.table_from_rows_device(S, G, C, rows.data_ptr(), lT.data_ptr(), stream)
    lo = rank * C
    x_p, x_v, x_a, p_f = (dev_t(a[:, lo:lo + C]) for a in (xp, xv, xa, pf))
    p_out = torch.empty((S, C, 45), dtype=torch.float64, device=dev)
    v_out, a_out = torch.empty_like(p_out), torch.empty_like(p_out)
    lT_next = torch.empty((S, 45, C), dtype=torch.float64, device=dev)
    lT_gath = torch.empty((G, S, 45, C), dtype=torch.float64, device=dev)
    status = torch.zeros((S, C), dtype=torch.int32, device=dev)
    info = torch.zeros((S, C, 8), dtype=torch.int32, device=dev)

    def one_step():
        if in_lib:     # solve of this rank's cluster + the all-gather of the new predictions, both enqueued by the library
            dmpc.step_sharded_device(S, N, lT.data_ptr(), x_p.data_ptr(), x_v.data_ptr(), x_a.data_ptr(), p_f.data_ptr(), p_out.data_ptr(),
                                     v_out.data_ptr(), a_out.data_ptr(), lT_gath.data_ptr(), status.data_ptr(), info.data_ptr(), stream)
            return
        dmpc.step_device(S, G, C, rank, lT.data_ptr(), x_p.data_ptr(), x_v.data_ptr(), x_a.data_ptr(), p_f.data_ptr(),
                         p_out.data_ptr(), v_out.data_ptr(), a_out.data_ptr(), lT_next.data_ptr(), status.data_ptr(),
                         info.data_ptr(), stream)
        if use_dist:   # the per-step exchange: every rank publishes its agents' new predictions
            if share:
                h_in = lT_next.cpu()
                h_out = torch.empty((G,) + tuple(h_in.shape), dtype=h_in.dtype)
                dist.all_gather_into_tensor(h_out.view(-1), h_in.view(-1))
                lT_gath.copy_(h_out)
            else:
                dist.all_gather_into_tensor(lT_gath.view(-1), lT_next.view(-1))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    dmpc.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    t1 = time.perf_counter()
    kern_ms, scan_ms, n_launch = dmpc.profile_read2()
    dmpc.profile(False)
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: the gathered table is the same on every rank and holds this rank's chunk in its slot
    exchange_ok = None
    if use_dist:
        ck = lT_gath.view(torch.int64).sum(dtype=torch.int64)    # bit-pattern checksum
        ck = torch.stack([ck, -ck]).to("cpu" if share else dev)
        dist.all_reduce(ck, op=dist.ReduceOp.MAX)
        if in_lib:   # this rank's slot must hold what a plain (unsharded-API) solve of its chunk writes
            dmpc.step_device(S, G, C, rank, lT.data_ptr(), x_p.data_ptr(), x_v.data_ptr(), x_a.data_ptr(), p_f.data_ptr(), p_out.data_ptr(),
                             v_out.data_ptr(), a_out.data_ptr(), lT_next.data_ptr(), status.data_ptr(), info.data_ptr(), stream)
            torch.cuda.synchronize()
        exchange_ok = bool(torch.equal(lT_gath[rank], lT_next)) and int(ck[0].item()) == -int(ck[1].item())
        # ... and the sharded step must be the UNSHARDED step of the same scenes, per agent and bit for bit: the first scenes solved
        # again as ONE chunk of all N agents on this GPU (table in the G = 1 layout), this rank's agents compared
        Sv = min(S, 8)
        rows1 = dev_t(l[:Sv]); lT1 = torch.empty((1, Sv, 45, N), dtype=torch.float64, device=dev)
        dmpc.table_from_rows_device(Sv, 1, N, rows1.data_ptr(), lT1.data_ptr(), stream)
        f1 = [dev_t(a_[:Sv]) for a_ in (xp, xv, xa, pf)]
        p1 = torch.empty((Sv, N, 45), dtype=torch.float64, device=dev); v1, a1 = torch.empty_like(p1), torch.empty_like(p1)
        st1 = torch.zeros((Sv, N), dtype=torch.int32, device=dev)
        ds = mp.Dmpc(cfg["variant"], device=local_rank, **kw)
        ds.step_device(Sv, 1, N, 0, lT1.data_ptr(), f1[0].data_ptr(), f1[1].data_ptr(), f1[2].data_ptr(), f1[3].data_ptr(), p1.data_ptr(), v1.data_ptr(),
                       a1.data_ptr(), 0, st1.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(p1[:, lo:lo + C], p_out[:Sv])) and bool(torch.equal(st1[:, lo:lo + C], status[:Sv]))
        okt = torch.tensor([1 if same else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        exchange_ok = exchange_ok and int(okt.item()) == 1
        del ds

    solves_per_step = S * N if not emu else S * C
    value = solves_per_step * args.steps / elapsed

    # secondary workloads (reported, not the headline)
    secondary = None
    if not args.no_secondary and G == 1 and rank == 0:
        secondary = []

        def replay(variant, what, cap_step):
            """steady-state replay of a captured MPC step of the same start/goal sets with another solver variant"""
            dv = mp.Dmpc(variant, device=local_rank, **kw)
            lv, xpv, xvv, xav, pfv, alivev = capture_state(dv, dict(cfg, variant=variant), S, N, cap_step, wl.SEED0 + 2)
            rowsv = dev_t(lv)
            lTv = torch.empty((1, S, 45, C), dtype=torch.float64, device=dev)
            dv.table_from_rows_device(S, 1, C, rowsv.data_ptr(), lTv.data_ptr(), stream)
            tv = [dev_t(a_) for a_ in (xpv, xvv, xav, pfv)]
            def stepv():
                dv.step_device(S, 1, C, 0, lTv.data_ptr(), tv[0].data_ptr(), tv[1].data_ptr(), tv[2].data_ptr(), tv[3].data_ptr(),
                               p_out.data_ptr(), v_out.data_ptr(), a_out.data_ptr(), lT_next.data_ptr(), status.data_ptr(),
                               info.data_ptr(), stream)
            for _ in range(3):
                stepv()
            torch.cuda.synchronize()
            tt = time.perf_counter()
            for _ in range(args.steps):
                stepv()
            torch.cuda.synchronize()
            el = time.perf_counter() - tt
            stv = status.cpu().numpy(); infv = info.cpu().numpy()
            secondary.append({"workload": f"{C} agents/scene, variant {variant} ({what}), {S} scenes, replay of MPC step {cap_step} "
                                          f"({int(alivev.sum())}/{S} scenes alive)",
                              "value": S * N * args.steps / el, "unit": "solves/s", "ms_per_step": el / args.steps * 1e3,
                              "solved_frac": float((stv & 1).mean()), "infeasible_frac": float(((stv & 8) != 0).mean()),
                              "mean_iters": float(infv[..., 4].mean()), "max_iters": int(infv[..., 4].max()), "max_tries": int(infv[..., 2].max()),
                              "invalid": int(((stv & 48) != 0).sum()), "mean_rows": float(infv[..., 1].mean())})
        # the reference's primary variant (test/failure_rate.m) at MPC step 12 of the live transitions, and the second C2 variant
        # SURVEY.md 8d names (test/comp_hardsoft2.m:247: solveHardDMPCOnDemand; like solveHardDMPC it dies at its first solve)
        replay("bound", "solveSoftDMPCbound, failure_rate.m constants", 12)
        replay("ondemand", "solveHardDMPCOnDemand", 12)
        # the same headline workload at a small batch: a 64-scene launch is bound by its slowest agent, not by throughput
        if S > 64:
            S3 = 64
            lT3 = torch.empty((1, S3, 45, C), dtype=torch.float64, device=dev)
            rows3 = rows[:S3].contiguous()
            dmpc.table_from_rows_device(S3, 1, C, rows3.data_ptr(), lT3.data_ptr(), stream)
            t3 = [x[:S3].contiguous() for x in (x_p, x_v, x_a, p_f)]
            def step3():
                dmpc.step_device(S3, 1, C, 0, lT3.data_ptr(), t3[0].data_ptr(), t3[1].data_ptr(), t3[2].data_ptr(), t3[3].data_ptr(),
                                 p_out.data_ptr(), v_out.data_ptr(), a_out.data_ptr(), lT_next.data_ptr(), status.data_ptr(),
                                 info.data_ptr(), stream)
            for _ in range(3):
                step3()
            torch.cuda.synchronize()
            tt = time.perf_counter()
            for _ in range(args.steps):
                step3()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - tt
            secondary.append({"workload": f"headline workload at {S3} scenes per step ({S3 * C} QPs per launch: latency-bound by the slowest agent)",
                              "value": S3 * C * args.steps / el3, "unit": "solves/s", "ms_per_step": el3 / args.steps * 1e3})
        # whole closed-loop transitions on the device (dmpc_transition): the quantity the reference's own recordings
        # report (MATLAB 63.6 s, C++/OOQP 12.4 s / 4.2 s with 1 / 8 threads per 100-agent transition)
        cfgT = dict(wl.CONFIGS["C4"])
        kwT = wl.solver_kwargs(cfgT, 100)
        dT = mp.Dmpc("bound", device=local_rank, **kwT)
        poT, pfT = wl.make_scenes(cfgT, 512, 100, wl.SEED0 + 100)
        dT.transition(poT[:1], pfT[:1], 10, cfgT["error_tol"])   # warm-up
        dT.transition(poT[:128], pfT[:128], 4, cfgT["error_tol"])   # ... and of the batch parts (their contexts are created on first use)
        for St in (1, 8, 512):
            tt = time.perf_counter()
            resT = dT.transition(poT[:St], pfT[:St], cfgT["K_T"], cfgT["error_tol"])
            dtT = time.perf_counter() - tt
            usedT = resT["K_T_used"]
            secondary.append({"workload": f"{St} whole transition(s), 100 agents, solveSoftDMPCbound (failure_rate.m constants), closed "
                                          f"loop on device incl. initDMPC, table swap, ReachedGoal and host<->device copies",
                              "wall_ms": dtT * 1e3, "ms_per_transition": dtT * 1e3 / St, "mpc_steps": ([int(u) for u in usedT] if St <= 8 else {"min": int(min(usedT)), "mean": float(sum(usedT)) / St, "max": int(max(usedT))}),
                              "completed": int(((resT["scene_status"] & 256) != 0).sum()),
                              "value": float(((usedT - 1) * 100).sum() / dtT), "unit": "solves/s"})
        # ONE scene -- the literal "100 agents" of BASELINE configs[1] / the reference's own use: a step is bound by the latency of
        # its slowest agent plus the launches.  Closed loop on the device (dmpc_transition), wall time per MPC step.
        for vname, cname in (("bound", "C4"), ("hard", "C2")):
            cfg1 = dict(wl.CONFIGS[cname]); kw1 = wl.solver_kwargs(cfg1, 100)
            d1 = mp.Dmpc(vname, device=local_rank, **kw1)
            po1, pf1 = wl.make_scenes(cfg1, 1, 100, wl.SEED0 + 100)
            d1.transition(po1, pf1, 10, cfg1["error_tol"], histories=False)
            best, res1 = 1e9, None
            for _ in range(5):
                tt = time.perf_counter(); res1 = d1.transition(po1, pf1, cfg1["K_T"], cfg1["error_tol"], histories=False); best = min(best, time.perf_counter() - tt)
            steps1 = max(int(res1["K_T_used"][0]) - 1, 1)
            secondary.append({"workload": f"ONE scene of 100 agents, variant {vname}, whole transition on the device: wall time per MPC step",
                              "us_per_mpc_step": best / steps1 * 1e6, "mpc_steps": steps1, "scene_status": int(res1["scene_status"][0]),
                              "value": 100 * steps1 / best, "unit": "solves/s"})
        # mixed precision (DMPC_PREC_MIXED: fp32 table, scan and rows; fp64 QP) against fp64 on whole transitions (BASELINE configs[4])
        for prec in ("f64", "mixed"):
            dm = mp.Dmpc("bound", device=local_rank, precision=prec, **kwT)
            dm.transition(poT[:8], pfT[:8], 10, cfgT["error_tol"], histories=False)
            dm.transition(poT[:128], pfT[:128], 4, cfgT["error_tol"], histories=False)   # (creates the contexts of the batch parts)
            dtm = 1e9
            for _ in range(3):   # best of three: the first full-size call also allocates the history buffers of the batch parts
                tt = time.perf_counter(); rm = dm.transition(poT, pfT, cfgT["K_T"], cfgT["error_tol"], histories=False); dtm = min(dtm, time.perf_counter() - tt)
            secondary.append({"workload": f"512 whole transitions, 100 agents, solveSoftDMPCbound, precision {prec}, histories left on the device (best of 3 calls)",
                              "wall_ms": dtm * 1e3, "completed": int(((rm["scene_status"] & 256) != 0).sum()),
                              "value": float(((rm["K_T_used"] - 1) * 100).sum() / dtm), "unit": "solves/s"})
        # BASELINE configs[3] (C4): ONE scene of 10^4 agents, solveSoftDMPCbound -- the first MPC steps of the closed loop on the device
        # (large scenes: neighbour pre-pass + list walk in the scan, crash start of the acceleration bounds in the solve)
        cfg4 = dict(wl.CONFIGS["C4"]); N4 = 10000
        kw4 = wl.solver_kwargs(cfg4, N4)
        po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)
        d4 = mp.Dmpc("bound", device=local_rank, **kw4)
        # (a whole transition stops at the first infeasible agent -- MPC step 4 of this scene; the steps are therefore driven from
        # the host here, failed agents keeping their previous prediction, and timed with the library's HIP events)
        l4, _, _ = d4.init_batch(po4, pf4)
        x4p, x4v, x4a = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
        sc4, so4, ok4 = [], [], []
        for k4 in range(9):
            d4.profile(True)
            o4 = d4.step_batch(l4, x4p, x4v, x4a, pf4)
            sms, cms, _ = d4.profile_read2()
            sc4.append(cms); so4.append(sms)
            okm = (o4["status"] == 1); ok4.append(float(okm.mean()))
            l4 = np.where(okm[..., None], o4["p"], l4); x4p = np.where(okm[..., None], o4["p"][..., :3], x4p)
            x4v = np.where(okm[..., None], o4["v"][..., :3], x4v); x4a = np.where(okm[..., None], o4["a"][..., :3], x4a)
        secondary.append({"workload": f"C4: ONE scene of {N4} agents, solveSoftDMPCbound, MPC steps 2-10 (device time per step: scan incl. neighbour "
                                      "pre-pass, solve incl. order and second tier; HIP events)",
                          "scan_ms": [round(x, 3) for x in sc4], "solve_ms": [round(x, 3) for x in so4], "solved_frac": ok4,
                          "ms_per_mpc_step": float(np.mean(sc4[1:]) + np.mean(so4[1:])), "first_step_ms": sc4[0] + so4[0],
                          "value": N4 / ((np.mean(sc4[1:]) + np.mean(so4[1:])) * 1e-3), "unit": "solves/s"})
        del d4
        # the same scene in mixed precision (fp32 table, scan and rows; fp64 QP): the O(N) part of the step at half the bytes
        d4m = mp.Dmpc("bound", device=local_rank, precision="mixed", **kw4)
        l4, _, _ = d4m.init_batch(po4, pf4)
        x4p, x4v, x4a = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
        scm, som = [], []
        for k4 in range(9):
            d4m.profile(True)
            o4 = d4m.step_batch(l4, x4p, x4v, x4a, pf4)
            sms, cms, _ = d4m.profile_read2()
            scm.append(cms); som.append(sms)
            okm = (o4["status"] == 1)
            l4 = np.where(okm[..., None], o4["p"], l4); x4p = np.where(okm[..., None], o4["p"][..., :3], x4p)
            x4v = np.where(okm[..., None], o4["v"][..., :3], x4v); x4a = np.where(okm[..., None], o4["a"][..., :3], x4a)
        secondary.append({"workload": f"C4 as above in mixed precision (DMPC_PREC_MIXED)", "scan_ms": [round(x, 3) for x in scm], "solve_ms": [round(x, 3) for x in som],
                          "ms_per_mpc_step": float(np.mean(scm[1:]) + np.mean(som[1:])), "value": N4 / ((np.mean(scm[1:]) + np.mean(som[1:])) * 1e-3), "unit": "solves/s"})
        del d4m
        # BASELINE configs[2] (C3: 1000 agents, solveSoftDMPC, test/success_test_softdmpc.m constants, randomExchange) and configs[4] (C5: 200
        # agents, solveSoftDMPCrepair, test/comp_repair.m constants): the first MPC steps of the closed loop, ONE scene (the reference's own
        # call pattern) and a Monte-Carlo batch of scenes; device time per step by the library's HIP events (scan + solve)
        def config_steps(cname, variant, Nc, Sc, nsteps=5):
            cfgc = dict(wl.CONFIGS[cname]); kwc = wl.solver_kwargs(cfgc, Nc)
            dc = mp.Dmpc(variant, device=local_rank, **kwc)
            poc, pfc = wl.make_scenes_device(dc, cfgc, Sc, Nc, wl.SEED0 + 7)
            lc, _, _ = dc.init_batch(poc, pfc)
            xp_, xv_, xa_ = poc.copy(), np.zeros_like(poc), np.zeros_like(poc)
            ms, okf, its = [], [], []
            for _ in range(nsteps):
                dc.profile(True)
                oc = dc.step_batch(lc, xp_, xv_, xa_, pfc)
                sms, cms, _ = dc.profile_read2()
                ms.append(sms + cms)
                okm = (oc["status"] == 1); okf.append(float(okm.mean())); its.append(float(oc["info"][..., 4].mean()))
                lc = np.where(okm[..., None], oc["p"], lc); xp_ = np.where(okm[..., None], oc["p"][..., :3], xp_)
                xv_ = np.where(okm[..., None], oc["v"][..., :3], xv_); xa_ = np.where(okm[..., None], oc["a"][..., :3], xa_)
            secondary.append({"workload": f"{cname}: {Sc} scene(s) of {Nc} agents, variant {variant}, MPC steps 2-{nsteps + 1} of the closed loop (device time per step: "
                                          "scan + solve, HIP events; failed agents keep their previous prediction)",
                              "step_ms": [round(x, 3) for x in ms], "ms_per_mpc_step": float(np.mean(ms)), "solved_frac": okf, "mean_iters": its,
                              "value": Sc * Nc / (float(np.mean(ms)) * 1e-3), "unit": "solves/s"})
        for cname, variant, Nc, Sc in (("C3", "softall", 1000, 1), ("C3", "softall", 1000, 16), ("C5", "repair", 200, 1), ("C5", "repair", 200, 64)):
            try:
                config_steps(cname, variant, Nc, Sc)
            except Exception as e:   # noqa: BLE001
                secondary.append({"workload": f"{cname} {variant} N={Nc} S={Sc}", "error": str(e)[:200]})
        # one process, every visible GPU (dmpc_create(.., DMPC_DEVICE_ALL, ..): threads + peer copies inside the library) -- the path a MEX /
        # C++ caller gets; only when this process sees more than one GPU
        if torch.cuda.device_count() > 1:
            try:
                dg = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kwT)
                dg.transition(poT[:8], pfT[:8], 10, cfgT["error_tol"], histories=False)
                tt = time.perf_counter(); rg = dg.transition(poT, pfT, cfgT["K_T"], cfgT["error_tol"], histories=False); dtg = time.perf_counter() - tt
                secondary.append({"workload": f"512 whole transitions, 100 agents, solveSoftDMPCbound, ONE process on {dg.n_devices} GPUs (DMPC_DEVICE_ALL: agents of "
                                              "every scene sharded over the GPUs, peer copies between MPC steps)", "n_gpus": dg.n_devices, "wall_ms": dtg * 1e3,
                                  "completed": int(((rg["scene_status"] & 256) != 0).sum()), "value": float(((rg["K_T_used"] - 1) * 100).sum() / dtg), "unit": "solves/s"})
                del dg
            except Exception as e:   # noqa: BLE001
                secondary.append({"workload": "one process on all GPUs (DMPC_DEVICE_ALL)", "error": str(e)[:200]})
        # restore the headline workload's last outputs for the statistics below
        one_step()
        torch.cuda.synchronize()
    # Strong scaling (multi-rank runs only): ONE C4 scene of 10^4 agents sharded over the G ranks (dmpc.cpp:1600-1625 clusters),
    # whole closed loop inside the library (dmpc_transition_sharded: scan + solve of the own cluster + RCCL all-gather per step).
    # Bounded by a timeout: a collective that never returns must not cost the headline line.
    strong = None
    # only when the headline's exchange was verified on this rank AND on every other one (all ranks take the same branch)
    # (round 4: whenever the library's communicator is up -- the same condition on every rank -- whatever the headline's exchange check
    # said; that verdict rides along in the record)
    run_strong = bool(use_dist and in_lib)
    if run_strong:
        okx = torch.tensor([1 if exchange_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okx, op=dist.ReduceOp.MIN)
        headline_exchange_ok = int(okx.item()) == 1
    if run_strong:
        import threading
        box = {}

        def c4_strong():
            try:
                cfg4 = dict(wl.CONFIGS["C4"]); N4 = 10000
                kw4 = wl.solver_kwargs(cfg4, N4)
                po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)   # deterministic: the same on every rank
                torch.cuda.set_device(local_rank)
                d4 = mp.Dmpc("bound", device=local_rank, **kw4)
                idt = torch.zeros(129, dtype=torch.uint8, device=dev)
                if rank == 0:
                    try:
                        idt[:128].copy_(torch.frombuffer(bytearray(mp.Dmpc.comm_unique_id()), dtype=torch.uint8)); idt[128] = 1
                    except Exception:   # noqa: BLE001
                        pass
                dist.broadcast(idt, src=0)
                if int(idt[128].item()) != 1:
                    raise RuntimeError("no RCCL id from the library")
                d4.comm_init(bytes(idt[:128].cpu().numpy().tobytes()), G, rank)
                d4.transition_sharded(po4, pf4, 3, cfg4["error_tol"], histories=False)
                dist.barrier(); torch.cuda.synchronize()
                tt = time.perf_counter()
                r4 = d4.transition_sharded(po4, pf4, 12, cfg4["error_tol"], histories=False)
                torch.cuda.synchronize(); dist.barrier()
                dt4 = time.perf_counter() - tt
                tmax = torch.tensor([dt4], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                st4 = max(int(r4["K_T_used"][0]) - 1, 1)
                box["r"] = {"workload": f"C4 strong scaling: ONE scene of {N4} agents sharded over {G} rank(s), solveSoftDMPCbound, first {st4} MPC "
                                        "steps inside the library (dmpc_transition_sharded: RCCL all-gather of the table per step)",
                            "n_gpus": G, "n_ranks_seen": d4.comm_size(), "headline_exchange_verified": headline_exchange_ok,
                            "ms_per_mpc_step": float(tmax.item()) * 1e3 / st4, "mpc_steps": st4,
                            "value": N4 * st4 / float(tmax.item()), "unit": "solves/s", "scaling": "strong"}
                d4.comm_destroy()
            except Exception as e:   # noqa: BLE001
                box["r"] = {"workload": "C4 strong scaling", "error": str(e)[:200]}

        th = threading.Thread(target=c4_strong, daemon=True)
        th.start(); th.join(timeout=180.0)
        strong = box.get("r", {"workload": "C4 strong scaling", "error": "timed out after 180 s"})
        strong_hung = th.is_alive()
    else:
        strong_hung = False

    # --emulate-gpus G: the single-process multi-GPU path (DMPC_DEVICE_ALL) with its G ranks emulated on this GPU -- whole transitions as ONE
    # group and as two groups side by side (the exchange of one half under the solve of the other).  The ranks share the device here, so this
    # shows the host-side cost of the exchange protocol (threads, barriers, events, same-device copies), not xGMI; never a headline number.
    group_overlap = None
    if emu and rank == 0:
        try:
            cfgE = dict(wl.CONFIGS["C4"]); NE, SE = 100 * emu, 128
            kwE = wl.solver_kwargs(cfgE, NE)
            poE, pfE = wl.make_scenes(cfgE, SE, NE, wl.SEED0 + 60)
            group_overlap = {"workload": f"{SE} scenes x {NE} agents, solveSoftDMPCbound, 40 MPC steps, ONE process, {emu} ranks emulated on one GPU"}
            for label, opts in (("one_group_ms_per_step", {"no_split": 1}), ("two_groups_ms_per_step", {})):
                mp.Dmpc.emulate_devices(emu)
                dg = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kwE)
                for k_, v_ in opts.items():
                    dg.debug_option(k_, v_)
                dg.transition(poE[:8], pfE[:8], 6, cfgE["error_tol"], histories=False)
                best = 1e9
                for _ in range(3):
                    tt = time.perf_counter(); rg = dg.transition(poE, pfE, 40, cfgE["error_tol"], histories=False); best = min(best, time.perf_counter() - tt)
                group_overlap[label] = best * 1e3 / max(int((rg["K_T_used"] - 1).max()), 1)
                del dg
        except Exception as e:   # noqa: BLE001
            group_overlap = {"error": str(e)[:200]}
        finally:
            mp.Dmpc.emulate_devices(0)

    st = status.cpu().numpy()
    inf = info.cpu().numpy()

    if rank == 0:
        # algorithmic HBM bytes per solve (SURVEY.md 8d / BASELINE.md 3): table read once per step by the
        # GPU (360 B per agent of the scene), state+goal 96 B, p/v/a horizons 1080 B, status 20 B
        b_alg = 360.0 * N / C + 1196.0
        achieved = (S * C) * b_alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # PMC counters cannot be read inside the timed run: traffic, issue fractions and the fp64 instruction classes come from the committed
        # rocprofv3 passes of this same workload (profiles/r03_pmc_summary.json, made by tools/gpu_profile_round.sh + tools/profile_summary.py)
        # -- and only when that summary was taken on THESE kernel sources (its source_hash: sha256 over multiagent_planning_amd/csrc/*.hip|*.h)
        traffic, pmc, pmc_note = None, None, None
        tj = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if os.path.exists(tj) and G == 1:
            try:
                jt = json.load(open(tj))
                if jt.get("solves_per_launch") != S * C:
                    pmc_note = f"profiles/{PMC_SUMMARY} is for another launch size: not used"
                elif jt.get("source_hash") != source_hash():
                    pmc_note = f"profiles/{PMC_SUMMARY} was measured on other kernel sources (hash {jt.get('source_hash')} != {source_hash()}): refused"
                else:
                    pmc = jt
                    traffic = jt.get("hbm_bytes_per_launch")
            except Exception as e:   # noqa: BLE001
                pmc, pmc_note = None, f"profiles/{PMC_SUMMARY}: {e}"
        # the box's practical HBM ceiling next to the 8 TB/s of the data sheet ("also measure a device-copy ceiling"): a 1 GiB
        # device-to-device copy, read + written bytes over HIP-event time, best of 5
        copy_gbs = None
        try:
            src_c = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst_c = torch.empty_like(src_c)
            dst_c.copy_(src_c); torch.cuda.synchronize()
            best_c = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dst_c.copy_(src_c); e1.record(); torch.cuda.synchronize()
                best_c = min(best_c, e0.elapsed_time(e1))
            copy_gbs = 2.0 * src_c.numel() * 4 / (best_c * 1e-3) / 1e9
            del src_c, dst_c
        except Exception:   # noqa: BLE001
            copy_gbs = None
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        kname = "dmpc_solve_persist_kernel" if S * C >= 16 * ncu * 8 else "dmpc_solve_kernel"
        line = {
            "metric": "agent-QP solves/sec (K=15 horizon)",
            "value": value, "unit": "solves/s", "n_gpus": (1 if emu else G), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" [SHARED-GPU FUNCTIONAL CHECK, not a measurement]" if share else ""),
            "config": {"workload": f"{args.config}: {C} agents/GPU x {G} GPU(s) per scene, variant {cfg['variant']} "
                                   f"(solve{'Hard' if cfg['variant']=='hard' else ''}DMPC), K=15, {S} scenes batched, "
                                   f"steady-state replay of the captured MPC step (requested {args.capture_step}; "
                                   f"{int(alive.sum())}/{S} scenes still alive there)",
                       "agents_per_scene": N, "scenes": S, "solves_per_step": solves_per_step,
                       "parallelism": f"agents sharded x{G}, all-gather per step" if G > 1 else "single GPU",
                       "exchange": exchange, "exchange_verified": exchange_ok, "n_ranks_seen": (dmpc.comm_size() if in_lib else (G if use_dist else 1)),
                       **({"debug_options": args.debug_option} if args.debug_option else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs if copy_gbs else None),
                         "kernel": kname, "kernel_ms_avg": kern_ms, "launches": n_launch,
                         "other_kernels_ms_avg": {"dmpc_scan_kernel+order_kernel": scan_ms},
                         "alg_bytes_per_solve": b_alg, "solves_per_launch": S * C,
                         "secondary": (dict(wave_time=pmc.get("wave_time"), instructions_per_solve=pmc.get("instructions_per_solve"),
                                            whole_step_traffic=pmc.get("whole_step_bytes_per_launch"), fp64=pmc.get("fp64"),
                                            traffic_calibration=pmc.get("traffic_calibration"), git_commit=pmc.get("git_commit"), source_hash=pmc.get("source_hash"),
                                            source=f"profiles/{PMC_SUMMARY} (rocprofv3 passes of this workload on these kernel sources, committed; not re-measured in this run)") if pmc else pmc_note),
                         "note": "on-chip bound (profiles/README.md): compulsory HBM traffic is ~1.5 KB/solve against thousands of dependent wave instructions, so the HBM "
                                 "fraction is small by construction; what bounds the kernel -- per-wave active / waiting cycles, VALU-pipe busy fraction, fp64 share -- is in "
                                 "`secondary` (counters, not a model).  No MFMA: the only dense contraction of the path, the Hessian / Gram build, does not depend on the "
                                 "inputs and is precomputed on the host (three 30x30 tables); what is left per solve are 8-wide triangular products of a factor that "
                                 "changes by one column per iteration -- 655 fp64 FMA of 3.7 k VALU instructions per solve (profiles/r04_pmc_sq3_*)"},
            # `value` counts every agent-step launched (one call of the reference's per-agent solver each); of those, the share that ended
            # with a solution / with a proof of infeasibility (agents the scan certifies infeasible never enter the solver):
            "value_solved_only": value * float((st & 1).mean()),
            "workload_stats": {"solved_frac": float((st & 1).mean()), "infeasible_frac": float(((st & 8) != 0).mean()),
                               "invalid": int(((st & 48) != 0).sum()), "mean_iters": float(inf[..., 4].mean()),
                               "max_iters": int(inf[..., 4].max()), "mean_rows": float(inf[..., 1].mean()),
                               "tries_histogram": {str(int(t)): int(c) for t, c in zip(*np.unique(inf[..., 2], return_counts=True))},
                               "max_rows": int(inf[..., 1].max()), "max_working_set": int(inf[..., 7].max()),
                               "scenes_alive_at_capture": int(alive.sum())},
        }
        if not args.no_secondary and G == 1:
            line["secondary"] = secondary
        if strong is not None:
            line["strong_scaling"] = strong
        if group_overlap is not None:
            line["single_process_group_emulated"] = group_overlap
        if not args.no_cpu_baseline and G == 1:
            # CPU baseline: the oracle (the literal dense QP of the .m files + dense Goldfarb-Idnani, oracle/dmpc_oracle.c) on
            # the same captured step, timed on this box's host cores.  The scenes of the batch are independent problems, so
            # the host runs them SCENE-parallel (one thread per scene, orc_step_scenes): T in {1, 2, 4, 8, physical cores},
            # ~2 s of wall time each on a bounded sample of whole scenes.  The reference's own threading -- 8 contiguous agent
            # clusters inside ONE scene (dmpc/cpp/dmpc.cpp:1600-1625) -- is reported next to it.
            from oracle import oracle as orc
            prm = orc.make_params(cfg["variant"], **kw)
            ncpu = os.cpu_count() or 1
            try:
                import psutil
                phys = psutil.cpu_count(logical=False) or ncpu
            except Exception:
                phys = ncpu
            sweep = []
            for T in sorted(set([1, 2, 4, 8, phys])):
                if T > max(phys, 1):
                    continue
                ns = min(S, 4 * T)
                sl = slice(0, ns)
                tt = time.perf_counter()
                reps = 0
                while reps == 0 or (time.perf_counter() - tt < 2.0 and reps < 64):
                    orc.step_scenes(prm, l[sl], xp[sl], xv[sl], xa[sl], pf[sl], nthreads=T)
                    reps += 1
                wall = time.perf_counter() - tt
                sweep.append({"threads": T, "value": reps * ns * N / wall, "scenes": ns, "repeats": reps, "wall_s": wall})
            tt = time.perf_counter()
            reps = 0
            while time.perf_counter() - tt < 2.0:
                orc.step(prm, l[reps % S], xp[reps % S], xv[reps % S], xa[reps % S], pf[reps % S], nthreads=min(8, N))
                reps += 1
            ref_style = reps * N / (time.perf_counter() - tt)
            best = max(sweep, key=lambda e: e["value"])
            line["cpu_baseline"] = {"value": best["value"], "unit": "solves/s", "cores": best["threads"], "kind": "port",
                                    "sample": f"{best['scenes']} scene(s) x {N} agents of the same captured step x {best['repeats']} repeat(s), "
                                              f"oracle/dmpc_oracle.c (literal dense QP + dense Goldfarb-Idnani), scene-parallel threads",
                                    "single_thread": sweep[0]["value"], "thread_sweep": sweep, "host_cpus": ncpu, "physical_cores": phys,
                                    "reference_style_8_clusters_in_one_scene": ref_style}
    if use_dist and not strong_hung:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: flush whatever the native libraries (RCCL's version banner) still
        # hold in C stdio buffers first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if strong_hung:   # a collective of the strong-scaling extra never returned: leave without waiting for its thread
        os._exit(0)


if __name__ == "__main__":
    main()
