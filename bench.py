#!/usr/bin/env python3
"""bench.py -- agent-QP solves/sec of the DMPC per-agent horizon QP (K = 15) on MI355X.

Contract (one JSON line on rank 0):  python bench.py --gpus N --steps K --warmup W

Headline = the configuration BASELINE.json's metric is quoted on ("at N agents, 1/2/4/8 MI355X"): configs[3] (C4), ONE scene of 10 000
agents, solveSoftDMPCbound with the constants of test/failure_rate.m, at EVERY N:
  * a "step" = ONE MPC step of the closed loop (dmpc_soft_bound.m:116-146 / dmpc.cpp:1656-1686): scan + rows + QP + propagate for every
    agent, the state advance, and the table swap `l = new_l`; table, states and goals resident in HBM, nothing crosses PCIe in the loop;
  * the steps are MPC steps 2-10 of the transition (step 1 is initDMPC), run again from the initDMPC state every nine steps, so that the
    workload of a step does not depend on how many steps the driver asks for; an agent whose QP fails keeps its state and prediction
    (the reference would abort the trial there; `workload_stats.solved_frac` says how many);
  * N = 1: dmpc_step_device; N > 1: one process per GPU, the SAME scene sharded in the reference's contiguous clusters
    (dmpc.cpp:1600-1625), dmpc_step_sharded_device = solve of the own cluster + ONE RCCL all-gather of the new predictions inside the
    library per step.  Total work is fixed -> "scaling": "strong"; the N = 1 point of a scaling curve IS this bench line;
  * value = agents x steps / max-over-ranks wall time between barriers.
`secondary` (N = 1 only) keeps the other configs and variants as replays / closed loops, each with its own fraction of the HBM roofline --
among them C2 x 512 scenes, the headline of rounds 1-4.
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
PMC_SUMMARY = "r06_c4_pmc_summary.json"
CYCLE = 9               # MPC steps 2 .. 10


def source_hash():
    """sha256 over the kernel sources (the key a committed counter summary must carry to be quoted next to a live measurement)"""
    import hashlib
    d = os.path.join(ROOT, "multiagent_planning_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def b_alg(N, N_loc):
    """algorithmic HBM bytes per solve (SURVEY.md 8d / BASELINE.md 3): table read once per step by the GPU (360 B per agent of the
    scene), state + goal 96 B, p/v/a horizons 1080 B, status 20 B"""
    return 360.0 * N / N_loc + 1196.0


def frac_1gpu(solves_per_s):
    return solves_per_s * b_alg(1, 1) / 1e9 / HBM_PEAK_GBS


def capture_state(dmpc, cfg, S, N, k_cap, seed):
    """Closed-loop run of S scenes up to MPC step k_cap on the GPU (host-array API); returns the table and states that are the INPUT
    of step k_cap+1.  Scenes that abort earlier (an agent infeasible / collided, the reference `break`s the trial) keep the last valid state."""
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


class ClosedLoop:
    """One scene batch in a device-resident closed loop: tables (double-buffered), states, outputs as torch tensors; one MPC step =
    the library's step entry point on device pointers + dmpc_advance_device + the table swap.  G > 1: this rank's cluster of every scene."""

    def __init__(self, torch, dmpc, dev, po, pf, G=1, rank=0, sharded=False):
        from multiagent_planning_amd import _lib
        self.torch, self.d, self.dev = torch, dmpc, dev
        self.S, self.N = po.shape[0], po.shape[1]
        self.G, self.rank, self.sharded = G, rank, sharded
        self.lo, self.cnt, self.cmax = _lib.partition(self.N, G, rank)
        if self.N % G:
            raise ValueError("the bench shards scenes whose agent count is a multiple of the rank count")
        S, N, cmax, cnt, lo = self.S, self.N, self.cmax, self.cnt, self.lo
        self.stream = torch.cuda.current_stream().cuda_stream
        t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
        l0, _, _ = dmpc.init_batch(po, pf)                                  # initDMPC.m: straight lines [S][N][45]
        rows = t(l0)
        self.lT0 = torch.empty((G, S, 45, cmax), dtype=torch.float64, device=dev)
        dmpc.table_from_rows_device(S, G, cmax, rows.data_ptr(), self.lT0.data_ptr(), self.stream)
        self.tab = [torch.empty_like(self.lT0), torch.empty_like(self.lT0)]
        own = slice(lo, lo + cnt)
        self.x0 = [t(po[:, own]), torch.zeros((S, cnt, 3), dtype=torch.float64, device=dev), torch.zeros((S, cnt, 3), dtype=torch.float64, device=dev)]
        self.x = [torch.empty_like(a) for a in self.x0]
        self.pf = t(pf[:, own])
        self.out = [torch.empty((S, cnt, 45), dtype=torch.float64, device=dev) for _ in range(3)]
        self.status = torch.zeros((S, cnt), dtype=torch.int32, device=dev)
        self.info = torch.zeros((S, cnt, 8), dtype=torch.int32, device=dev)
        self.cur = 0
        self.k = 0          # MPC steps done since the last reset
        self.reset()

    def reset(self):
        self.tab[self.cur].copy_(self.lT0)
        if self.G > 1 and not self.sharded:   # (no exchange: the chunks of the other ranks are never written)
            self.tab[self.cur ^ 1].copy_(self.lT0)
        for a, b in zip(self.x, self.x0):
            a.copy_(b)
        self.k = 0

    def step(self):
        d, S = self.d, self.S
        cur, nxt = self.tab[self.cur], self.tab[self.cur ^ 1]
        x, o = self.x, self.out
        if self.sharded:    # solve of this rank's cluster + the all-gather of the new predictions into the next table, both enqueued by the library
            d.step_sharded_device(S, self.N, cur.data_ptr(), x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), self.pf.data_ptr(), o[0].data_ptr(),
                                  o[1].data_ptr(), o[2].data_ptr(), nxt.data_ptr(), self.status.data_ptr(), self.info.data_ptr(), self.stream)
        else:               # one chunk: the kernel's next-table chunk [S][45][C] IS the next table
            d.step_device(S, self.G, self.cmax, self.rank, cur.data_ptr(), x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), self.pf.data_ptr(),
                          o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), nxt.data_ptr(), self.status.data_ptr(), self.info.data_ptr(), self.stream)
        d.advance_device(S * self.cnt, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), self.status.data_ptr(), x[0].data_ptr(), x[1].data_ptr(),
                         x[2].data_ptr(), self.stream)
        self.cur ^= 1
        self.k += 1

    def cycle_step(self, cycle=CYCLE):
        """one step of the repeating loop over MPC steps 2 .. cycle+1 (the reset of the state is part of the loop)"""
        if self.k >= cycle:
            self.reset()
        self.step()


def timed(torch, fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=27)
    ap.add_argument("--warmup", type=int, default=9)
    ap.add_argument("--agents", type=int, default=10000, help="agents of the headline scene (BASELINE configs[3]: 10 000)")
    ap.add_argument("--scenes", type=int, default=512, help="scenes batched by the secondary replays (512 ~ one Monte-Carlo experiment of the reference)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-gpus", type=int, default=0,
                    help="tuning aid: on ONE GPU run rank 0's share of a G-GPU job (no collective) to see the per-rank step time; never a headline number")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="development: a dmpc_debug_option of the headline context (A/B runs); recorded in the line")
    args = ap.parse_args()
    t_start = time.perf_counter()

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
    if world != args.gpus and world == 1 and args.gpus > 1 and not args.emulate_gpus:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    G = world
    emu = args.emulate_gpus if (world == 1 and args.emulate_gpus > 1) else 0
    if emu:
        G = emu
    # functional check of the multi-rank path on a box with ONE GPU (tests only, never a reported number): all ranks share cuda:0 and
    # the exchange is staged through the host (gloo)
    share = bool(os.environ.get("DMPC_BENCH_SHARE_GPU")) and world > 1
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = (G > 1 and not emu) or (bool(os.environ.get("DMPC_BENCH_FORCE_DIST")) and not emu)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=G)
        else:   # device_id binds the RCCL communicator to this rank's GPU up front (no device guessing in barrier())
            dist.init_process_group("nccl", rank=rank, world_size=G, device_id=dev)

    # ---------------------------------------------------------------- the headline scene
    cfg4 = dict(wl.CONFIGS["C4"])
    N4 = args.agents - args.agents % G     # (a multiple of the rank count: equal clusters)
    kw4 = wl.solver_kwargs(cfg4, N4)
    po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)   # deterministic: the same on every rank
    dmpc = mp.Dmpc("bound", device=local_rank, **kw4)
    for kv in args.debug_option:
        dmpc.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    # the per-step exchange runs INSIDE the library (dmpc_step_sharded_device: solve + ncclAllGather on one stream); the RCCL id of the
    # library's communicator travels through torch.distributed
    exchange = "none"
    if use_dist and not share:
        idt = torch.zeros(129, dtype=torch.uint8, device=dev)      # 128-byte id + "rank 0 has one" flag
        if rank == 0:
            try:
                idt[:128].copy_(torch.frombuffer(bytearray(mp.Dmpc.comm_unique_id()), dtype=torch.uint8))
                idt[128] = 1
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] rank 0: no RCCL id from the library ({e})\n")
        dist.broadcast(idt, src=0)
        okf = torch.zeros(1, dtype=torch.int32, device=dev)
        if int(idt[128].item()) == 1:      # the same on every rank: all of them enter comm_init (a collective) or none
            try:
                dmpc.comm_init(bytes(idt[:128].cpu().numpy().tobytes()), G, rank)
                okf = torch.ones(1, dtype=torch.int32, device=dev)
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] rank {rank}: library communicator unavailable ({e})\n")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0:
            sys.exit("bench.py: the library's RCCL communicator could not be set up on every rank")
        exchange = "in-library RCCL all-gather (dmpc_step_sharded_device)"
    elif share:
        exchange = "gloo through the host (shared-GPU functional check)"

    if share:
        # (all ranks on one GPU: the solve of every rank's cluster through dmpc_step_device, the exchange through the host -- the protocol of
        # the sharded loop without a device transport)
        loop = ClosedLoop(torch, dmpc, dev, po4, pf4, G, rank, sharded=False)
        nxt_chunk = torch.empty((1, 45, loop.cmax), dtype=torch.float64, device=dev)

        def one_step():
            if loop.k >= CYCLE:
                loop.reset()
            cur, nxt = loop.tab[loop.cur], loop.tab[loop.cur ^ 1]
            x, o = loop.x, loop.out
            dmpc.step_device(1, G, loop.cmax, rank, cur.data_ptr(), x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), loop.pf.data_ptr(), o[0].data_ptr(),
                             o[1].data_ptr(), o[2].data_ptr(), nxt_chunk.data_ptr(), loop.status.data_ptr(), loop.info.data_ptr(), loop.stream)
            dmpc.advance_device(loop.cnt, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), loop.status.data_ptr(), x[0].data_ptr(), x[1].data_ptr(),
                                x[2].data_ptr(), loop.stream)
            h_in = nxt_chunk.cpu()
            h_out = torch.empty((G,) + tuple(h_in.shape), dtype=h_in.dtype)
            dist.all_gather_into_tensor(h_out.view(-1), h_in.view(-1))
            nxt.copy_(h_out.view_as(nxt))
            loop.cur ^= 1; loop.k += 1
    else:
        loop = ClosedLoop(torch, dmpc, dev, po4, pf4, G, rank if not emu else 0, sharded=bool(use_dist))
        one_step = loop.cycle_step   # (emulated: rank 0's cluster of a G-rank job without the exchange, the other clusters keep their initDMPC predictions)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # workload statistics of the cycle (untimed): per MPC step the solved share and the iteration counts
    loop.reset()
    stats = []
    for _ in range(CYCLE):
        one_step()
        torch.cuda.synchronize()
        st = loop.status.cpu().numpy(); inf = loop.info.cpu().numpy()
        stats.append((float((st & 1).mean()), float(inf[..., 4].mean()), int(inf[..., 4].max()), int(inf[..., 2].max()), int(((st & 48) != 0).sum()),
                      float(inf[..., 1].mean()), int(inf[..., 7].max()), float(((st & 8) != 0).mean())))
    st_last = loop.status.clone()
    setup_s = time.perf_counter() - t_start

    loop.reset()
    for _ in range(args.warmup):
        one_step()
    loop.reset()          # the timed steps start at MPC step 2
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
        t = torch.tensor([elapsed, kern_ms, scan_ms], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_max, scan_max = (float(v) for v in t.tolist())
    else:
        kern_max, scan_max = kern_ms, scan_ms

    # outside the timed region: the sharded step must be the UNSHARDED step of the same scene, per agent and bit for bit -- the scene solved
    # once more as ONE chunk of all N agents on this GPU from the initDMPC state, this rank's agents compared; and every rank must hold the
    # same gathered table
    exchange_ok = None
    if use_dist:
        loop.reset(); one_step(); torch.cuda.synchronize()
        ds = mp.Dmpc("bound", device=local_rank, **kw4)
        one = ClosedLoop(torch, ds, dev, po4, pf4, 1, 0, sharded=False)
        one.step(); torch.cuda.synchronize()
        own = slice(loop.lo, loop.lo + loop.cnt)
        same = bool(torch.equal(one.out[0][:, own], loop.out[0])) and bool(torch.equal(one.status[:, own], loop.status))
        tab = loop.tab[loop.cur]
        same = same and bool(torch.equal(tab[rank][..., :loop.cnt], one.tab[one.cur][0][..., own]))
        ck = tab.view(torch.int64).sum(dtype=torch.int64)    # bit-pattern checksum of the whole gathered table
        ck = torch.stack([ck, -ck, torch.tensor(0 if same else 1, device=dev, dtype=torch.int64)]).to("cpu" if share else dev)
        dist.all_reduce(ck, op=dist.ReduceOp.MAX)
        exchange_ok = int(ck[0].item()) == -int(ck[1].item()) and int(ck[2].item()) == 0
        del one, ds

    value = N4 * args.steps / elapsed if not emu else loop.cnt * args.steps / elapsed

    # ---------------------------------------------------------------- secondary workloads (N = 1; reported, not the headline)
    secondary = None
    if not args.no_secondary and G == 1 and rank == 0:
        secondary = secondary_workloads(torch, mp, wl, dev, local_rank, args)

    if rank == 0:
        N_loc = loop.cnt
        ba = b_alg(N4, N_loc)
        achieved = N_loc * ba / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # PMC counters cannot be read inside the timed run: traffic and the per-wave counters come from the committed rocprofv3 passes of this
        # same workload (profiles/r05_c4_pmc_summary.json, made by tools/gpu_profile_round.sh + tools/profile_summary.py) -- and only when that
        # summary was taken on THESE kernel sources (its source_hash: sha256 over multiagent_planning_amd/csrc/*.hip|*.h)
        traffic, pmc, pmc_note = None, None, None
        tj = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if os.path.exists(tj) and G == 1:
            try:
                jt = json.load(open(tj))
                if jt.get("solves_per_launch") != N_loc:
                    pmc_note = f"profiles/{PMC_SUMMARY} is for another launch size: not used"
                elif jt.get("source_hash") != source_hash():
                    pmc_note = f"profiles/{PMC_SUMMARY} was measured on other kernel sources (hash {jt.get('source_hash')} != {source_hash()}): refused"
                else:
                    pmc = jt
                    traffic = jt.get("hbm_bytes_per_launch")
            except Exception as e:   # noqa: BLE001
                pmc, pmc_note = None, f"profiles/{PMC_SUMMARY}: {e}"
        # the box's practical HBM ceiling next to the 8 TB/s of the data sheet: a 1 GiB device-to-device copy, best of 5
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
        ms_step = elapsed / args.steps * 1e3
        sa = np.array([s[:2] for s in stats])
        stl = st_last.cpu().numpy()
        line = {
            "metric": "agent-QP solves/sec (K=15 horizon)",
            "value": value, "unit": "solves/s", "n_gpus": (1 if emu else G), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" [SHARED-GPU FUNCTIONAL CHECK, not a measurement]" if share else ""),
            "config": {"workload": f"C4 (BASELINE configs[3]): ONE scene of {N4} agents, solveSoftDMPCbound (test/failure_rate.m constants), K=15, first `coll` return at MPC step 4 (value counts such early returns as solves: value_solved_only), "
                                   f"device-resident closed loop over MPC steps 2-{CYCLE + 1} (from the initDMPC state again every {CYCLE} steps), "
                                   f"agents sharded over {G} GPU(s)" + (" [rank 0's cluster only, emulated]" if emu else ""),
                       "agents_per_scene": N4, "scenes": 1, "solves_per_step": (N4 if not emu else loop.cnt), "agents_per_gpu": N_loc,
                       "parallelism": (f"agents sharded x{G} in contiguous clusters (dmpc.cpp:1600-1625), one all-gather of the new predictions per MPC step" if G > 1 else "single GPU"),
                       "exchange": exchange, "exchange_verified": exchange_ok, "n_ranks_seen": (dmpc.comm_size() if (use_dist and not share) else (G if use_dist else 1)),
                       "setup_s": setup_s,
                       **({"debug_options": args.debug_option} if args.debug_option else {})},
            # where a step's wall time goes on the slowest rank: the library's HIP events around scan (+ neighbour lists, order) and solve, and
            # what is left -- the exchange (N > 1), the state advance and the launch gaps
            "step_breakdown_ms": {"scan_lists_order_events": scan_max, "solve_events": kern_max,
                                  "exchange_advance_and_gaps": ms_step - scan_max - kern_max},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs if copy_gbs else None),
                         # (the library names the kernel its last step launched for the bulk of the agents: dmpc_last_solve_kernel, ABI revision 7)
                         "kernel": dmpc.last_solve_kernel,
                         "kernel_ms_avg": kern_ms, "launches": n_launch,
                         "other_kernels_ms_avg": {"neighbour lists (bbox, table copy, grid_bin/scan/fill/query) + dmpc_scan_kernel + order_kernel": scan_ms},
                         "alg_bytes_per_solve": ba, "solves_per_launch": N_loc,
                         "whole_step_frac": (N_loc * ba / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS),
                         "secondary": (dict(wave_time=pmc.get("wave_time"), instructions_per_solve=pmc.get("instructions_per_solve"),
                                            whole_step_traffic=pmc.get("whole_step_bytes_per_launch"), fp64=pmc.get("fp64"),
                                            traffic_calibration=pmc.get("traffic_calibration"), git_commit=pmc.get("git_commit"), source_hash=pmc.get("source_hash"),
                                            source=f"profiles/{PMC_SUMMARY} (rocprofv3 passes of this workload on these kernel sources, committed; not re-measured in this run)") if pmc else pmc_note),
                         "note": "on-chip bound (profiles/README.md): compulsory HBM traffic is ~1.5 KB/solve against ~10 k dependent wave instructions, so the "
                                 "HBM fraction is small by construction; what bounds the kernel -- per-wave active / waiting cycles, VALU-pipe busy fraction -- is in `secondary` "
                                 "(counters, not a model).  No MFMA: the reduced solver of solveSoftDMPCbound (dmpc_rsolve.hip, round 6) has no dense contraction at all -- "
                                 "per axis a tridiagonal + rank-1 Hessian (parallel cyclic reduction in DPP rows + Sherman-Morrison), the soft rows as a 3x3 penalty, "
                                 "hard rows / walls / entering constraint as a system of at most eight unknowns, one per lane"},
            "value_solved_only": value * float(sa[:, 0].mean()),
            "workload_stats": {"per_mpc_step": [{"mpc_step": i + 2, "solved_frac": s[0], "mean_iters": s[1], "max_iters": s[2], "max_tries": s[3]} for i, s in enumerate(stats)],
                               "solved_frac": float(sa[:, 0].mean()), "mean_iters": float(sa[:, 1].mean()), "max_iters": int(max(s[2] for s in stats)),
                               "invalid": int(sum(s[4] for s in stats)), "mean_rows": float(np.mean([s[5] for s in stats])),
                               "max_working_set": int(max(s[6] for s in stats)), "infeasible_frac": float(np.mean([s[7] for s in stats])),
                               "last_step_status_histogram": {str(int(v)): int(c) for v, c in zip(*np.unique(stl, return_counts=True))}},
        }
        if secondary is not None:
            line["secondary"] = secondary
        if not args.no_cpu_baseline and G == 1:
            line["cpu_baseline"] = cpu_baseline(wl, cfg4, kw4, N4, po4, pf4, dmpc)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: flush whatever the native libraries (RCCL's version banner) still hold in C stdio buffers first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


def cpu_baseline(wl, cfg4, kw4, N4, po4, pf4, dmpc):
    """The oracle (the literal dense QP of the .m files + dense Goldfarb-Idnani, oracle/dmpc_oracle.c -- test infrastructure, used here only as
    the timed CPU baseline) on the SAME scene, timed on this box's host cores on a bounded sample: MPC steps 3, 6 and 10 of the headline's
    closed loop (states and tables produced by the GPU path through the host API), every agent of the scene scanning all 10 000 neighbours as
    CheckCollSoftDMPC.m:7-10 does.  Threads = the reference's own parallelisation: contiguous agent clusters inside ONE scene
    (dmpc.cpp:1600-1625), T in {8, physical cores}; T = 1 on a slice of the agents.  The library is compiled HERE for this host
    (-O3 -march=native; the checker the tests use is the portable x86-64-v3 build that travels with the repository)."""
    import tempfile
    from oracle import oracle as orc
    native = None
    try:
        native = orc.lib(orc.build(force=True, cflags=["-O3", "-march=native"], out=os.path.join(tempfile.mkdtemp(prefix="orc_native_"), "libdmpc_oracle_native.so")))
    except Exception:   # noqa: BLE001 -- no compiler on the box: the portable build
        native = None
    prm = orc.make_params("bound", **kw4)
    l, _, _ = dmpc.init_batch(po4, pf4)
    xp, xv, xa = po4.copy(), np.zeros_like(po4), np.zeros_like(po4)
    states = {}
    for k in range(2, 10):                                 # MPC steps 2 .. 9 on the GPU: the inputs of steps 3 .. 10
        out = dmpc.step_batch(l, xp, xv, xa, pf4)
        ok = (out["status"] & 1) == 1
        l = np.where(ok[..., None], out["p"], l); xp = np.where(ok[..., None], out["p"][..., :3], xp)
        xv = np.where(ok[..., None], out["v"][..., :3], xv); xa = np.where(ok[..., None], out["a"][..., :3], xa)
        if k + 1 in (3, 6, 10):
            states[k + 1] = (l[0].copy(), xp[0].copy(), xv[0].copy(), xa[0].copy())
    pf = pf4[0]
    ncpu = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or ncpu
    except Exception:
        phys = ncpu
    sweep = []
    l3, xp3, xv3, xa3 = states[3]
    n1 = min(N4, 400)
    tt = time.perf_counter()
    for n in range(n1):
        orc.solve_one(prm, l3, n, xp3[n], xv3[n], xa3[n], pf[n])
    w1 = time.perf_counter() - tt
    sweep.append({"threads": 1, "value": n1 / w1, "agents": n1, "wall_s": w1, "mpc_steps": [3], "build": "x86-64-v3"})
    for T in sorted(set([8, phys])):
        per_step = {}
        tt_all = time.perf_counter(); reps_all = 0
        for k, (lk, xpk, xvk, xak) in sorted(states.items()):
            tt = time.perf_counter(); reps = 0
            while reps == 0 or (time.perf_counter() - tt < 1.5 and reps < 8):
                orc.step(prm, lk, xpk, xvk, xak, pf, nthreads=T, library=native); reps += 1
            per_step[k] = reps * N4 / (time.perf_counter() - tt)
            reps_all += reps
        w = time.perf_counter() - tt_all
        sweep.append({"threads": T, "value": reps_all * N4 / w, "agents": N4, "repeats": reps_all, "wall_s": w, "mpc_steps": sorted(states), "per_mpc_step": per_step,
                      "build": "-march=native" if native is not None else "x86-64-v3"})
    best = max(sweep, key=lambda e: e["value"])
    return {"value": best["value"], "unit": "solves/s", "cores": best["threads"], "kind": "port",
            "sample": f"MPC steps 3, 6 and 10 of the headline scene ({N4} agents, all of them) x {best.get('repeats', 1)} pass(es) in all, oracle/dmpc_oracle.c (literal dense QP + dense "
                      "Goldfarb-Idnani; every agent scans the whole table), compiled on this host with -O3 -march=native, agents in contiguous clusters over the threads as dmpc.cpp:1600-1625",
            "single_thread": sweep[0]["value"], "thread_sweep": sweep, "host_cpus": ncpu, "physical_cores": phys}


def secondary_workloads(torch, mp, wl, dev, local_rank, args):
    """The other configs / variants on one GPU.  Every entry carries `roofline_frac` = its solves/s x 1556 B / 8 TB/s (one GPU holds the whole
    scene).  Keys say what was timed: `wall_*` = host wall clock around device-resident loops (launch gaps included), `device_events_*` = sums of
    the library's HIP-event intervals (scan + solve kernels only; host-API copies around them are NOT in it)."""
    sec = []
    S, C = args.scenes, 100
    stream = torch.cuda.current_stream().cuda_stream
    steps = max(5, min(args.steps, 20))
    dev_t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
    cfg2 = wl.CONFIGS["C2"]; kw2 = wl.solver_kwargs(cfg2, C)
    p_out = torch.empty((S, C, 45), dtype=torch.float64, device=dev); v_out, a_out = torch.empty_like(p_out), torch.empty_like(p_out)
    lT_next = torch.empty((S, 45, C), dtype=torch.float64, device=dev)
    status = torch.zeros((S, C), dtype=torch.int32, device=dev); info = torch.zeros((S, C, 8), dtype=torch.int32, device=dev)

    def add(entry):
        if "value" in entry and entry.get("unit") == "solves/s":
            entry["roofline_frac"] = frac_1gpu(entry["value"])
        sec.append(entry)

    def replay(variant, what, cap_step, Sr=S):
        """steady-state replay of a captured MPC step of the C2 start/goal sets (100 agents per scene) with a solver variant"""
        dv = mp.Dmpc(variant, device=local_rank, **kw2)
        lv, xpv, xvv, xav, pfv, alivev = capture_state(dv, dict(cfg2, variant=variant), Sr, C, cap_step, wl.SEED0 + 2)
        lTv = torch.empty((1, Sr, 45, C), dtype=torch.float64, device=dev)
        rowsv = dev_t(lv)
        dv.table_from_rows_device(Sr, 1, C, rowsv.data_ptr(), lTv.data_ptr(), stream)
        tv = [dev_t(a_) for a_ in (xpv, xvv, xav, pfv)]

        def stepv():
            dv.step_device(Sr, 1, C, 0, lTv.data_ptr(), tv[0].data_ptr(), tv[1].data_ptr(), tv[2].data_ptr(), tv[3].data_ptr(),
                           p_out.data_ptr(), v_out.data_ptr(), a_out.data_ptr(), lT_next.data_ptr(), status.data_ptr(), info.data_ptr(), stream)
        dv.profile(True)
        el = timed(torch, stepv, steps)
        sv, sc, _ = dv.profile_read2()
        stv = status[:Sr].cpu().numpy(); infv = info[:Sr].cpu().numpy()
        add({"workload": f"{C} agents/scene, variant {variant} ({what}), {Sr} scenes, steady-state replay of MPC step {cap_step} "
                         f"({int(alivev.sum())}/{Sr} scenes alive)", "timed": "wall clock, device-resident replay",
             "value": Sr * C * steps / el, "unit": "solves/s", "wall_ms_per_step": el / steps * 1e3, "device_events_solve_ms": sv, "device_events_scan_ms": sc,
             "solved_frac": float((stv & 1).mean()), "infeasible_frac": float(((stv & 8) != 0).mean()),
             "mean_iters": float(infv[..., 4].mean()), "max_iters": int(infv[..., 4].max()), "max_tries": int(infv[..., 2].max()),
             "invalid": int(((stv & 48) != 0).sum()), "mean_rows": float(infv[..., 1].mean())})

    # BASELINE configs[1] (C2, the headline of rounds 1-4): hard ellipsoidal rows; every hard-constrained scene dies at its first solve, as in the reference
    replay("hard", "solveHardDMPC, test/comp_hardsoft2.m constants: BASELINE configs[1]", 12)
    replay("hard", "solveHardDMPC at a small batch: bound by the slowest agent", 12, 64)
    replay("ondemand", "solveHardDMPCOnDemand", 12)
    # the reference's primary variant and its siblings at MPC step 12 of the live transitions
    replay("bound", "solveSoftDMPCbound, failure_rate.m constants", 12)
    replay("all3", "solveSoftDMPCall", 12)
    # whole closed-loop transitions on the device (dmpc_transition): the quantity the reference's own recordings report (MATLAB 63.6 s,
    # C++/OOQP 12.4 s / 4.2 s with 1 / 8 threads per 100-agent transition)
    cfgT = dict(wl.CONFIGS["C4"])
    kwT = wl.solver_kwargs(cfgT, 100)
    dT = mp.Dmpc("bound", device=local_rank, **kwT)
    poT, pfT = wl.make_scenes(cfgT, 512, 100, wl.SEED0 + 100)
    dT.transition(poT[:1], pfT[:1], 10, cfgT["error_tol"])   # warm-up
    dT.transition(poT[:128], pfT[:128], 4, cfgT["error_tol"])   # ... and of the batch parts (their contexts are created on first use)
    for St in (1, 512):
        tt = time.perf_counter()
        resT = dT.transition(poT[:St], pfT[:St], cfgT["K_T"], cfgT["error_tol"], histories=(St == 1))
        dtT = time.perf_counter() - tt
        usedT = resT["K_T_used"]
        add({"workload": f"{St} whole transition(s), 100 agents, solveSoftDMPCbound (failure_rate.m constants), closed loop inside the library incl. "
                         "initDMPC, table swap, ReachedGoal" + (", histories copied to the host" if St == 1 else ", histories left on the device"),
             "timed": "wall clock of dmpc_transition", "wall_ms": dtT * 1e3, "wall_ms_per_transition": dtT * 1e3 / St,
             "mpc_steps": ([int(u) for u in usedT] if St <= 8 else {"min": int(min(usedT)), "mean": float(sum(usedT)) / St, "max": int(max(usedT))}),
             "completed": int(((resT["scene_status"] & 256) != 0).sum()), "value": float(((usedT - 1) * 100).sum() / dtT), "unit": "solves/s"})
    # ONE scene -- the literal "100 agents" of BASELINE configs[1] / the reference's own use: a step is bound by the latency of its slowest agent
    for vname, cname in (("bound", "C4"), ("hard", "C2")):
        cfg1 = dict(wl.CONFIGS[cname]); kw1 = wl.solver_kwargs(cfg1, 100)
        d1 = mp.Dmpc(vname, device=local_rank, **kw1)
        po1, pf1 = wl.make_scenes(cfg1, 1, 100, wl.SEED0 + 100)
        d1.transition(po1, pf1, 10, cfg1["error_tol"], histories=False)
        best, res1 = 1e9, None
        for _ in range(5):
            tt = time.perf_counter(); res1 = d1.transition(po1, pf1, cfg1["K_T"], cfg1["error_tol"], histories=False); best = min(best, time.perf_counter() - tt)
        steps1 = max(int(res1["K_T_used"][0]) - 1, 1)
        add({"workload": f"ONE scene of 100 agents, variant {vname}, whole transition inside the library", "timed": "wall clock of dmpc_transition, best of 5",
             "wall_us_per_mpc_step": best / steps1 * 1e6, "mpc_steps": steps1, "scene_status": int(res1["scene_status"][0]),
             "value": 100 * steps1 / best, "unit": "solves/s"})
    # mixed precision (DMPC_PREC_MIXED: fp32 table, scan and rows; fp64 QP) against fp64 on whole transitions (BASELINE configs[4])
    dm = mp.Dmpc("bound", device=local_rank, precision="mixed", **kwT)
    dm.transition(poT[:8], pfT[:8], 10, cfgT["error_tol"], histories=False)
    dm.transition(poT[:128], pfT[:128], 4, cfgT["error_tol"], histories=False)
    dtm = 1e9
    for _ in range(2):
        tt = time.perf_counter(); rm = dm.transition(poT, pfT, cfgT["K_T"], cfgT["error_tol"], histories=False); dtm = min(dtm, time.perf_counter() - tt)
    add({"workload": "512 whole transitions, 100 agents, solveSoftDMPCbound, precision mixed (fp32 table / scan / rows, fp64 QP), histories left on the device",
         "timed": "wall clock of dmpc_transition, best of 2", "wall_ms": dtm * 1e3, "completed": int(((rm["scene_status"] & 256) != 0).sum()),
         "value": float(((rm["K_T_used"] - 1) * 100).sum() / dtm), "unit": "solves/s"})
    del dm, dT

    # BASELINE configs[2] (C3: 1000 agents, solveSoftDMPC, test/success_test_softdmpc.m constants, randomExchange) and configs[4] (C5: 200 agents,
    # solveSoftDMPCrepair, test/comp_repair.m constants): device-resident closed loops over MPC steps 2-6, ONE scene (the reference's own call
    # pattern) and a Monte-Carlo batch of scenes
    def config_loop(cname, variant, Nc, Sc, nsteps=5):
        cfgc = dict(wl.CONFIGS[cname]); kwc = wl.solver_kwargs(cfgc, Nc)
        dc = mp.Dmpc(variant, device=local_rank, **kwc)
        poc, pfc = wl.make_scenes_device(dc, cfgc, Sc, Nc, wl.SEED0 + 7)
        lp = ClosedLoop(torch, dc, dev, poc, pfc)
        okf, its = [], []
        for _ in range(nsteps):
            lp.step(); torch.cuda.synchronize()
            okf.append(float((lp.status.cpu().numpy() & 1).mean())); its.append(float(lp.info.cpu().numpy()[..., 4].mean()))
        reps = 3
        dc.profile(True)
        torch.cuda.synchronize()
        tt = time.perf_counter()
        for _ in range(reps):
            lp.reset()
            for _ in range(nsteps):
                lp.step()
        torch.cuda.synchronize()
        el = time.perf_counter() - tt
        sv, sc, _ = dc.profile_read2()
        add({"workload": f"{cname}: {Sc} scene(s) of {Nc} agents, variant {variant}, device-resident closed loop over MPC steps 2-{nsteps + 1} (failed agents keep their previous prediction)",
             "timed": "wall clock, device-resident loop", "wall_ms_per_mpc_step": el / (reps * nsteps) * 1e3, "device_events_solve_ms": sv, "device_events_scan_ms": sc,
             "solved_frac": okf, "mean_iters": its, "value": Sc * Nc * reps * nsteps / el, "unit": "solves/s"})
    for cname, variant, Nc, Sc in (("C3", "softall", 1000, 1), ("C3", "softall", 1000, 16), ("C5", "repair", 200, 1), ("C5", "repair", 200, 64)):
        try:
            config_loop(cname, variant, Nc, Sc)
        except Exception as e:   # noqa: BLE001
            sec.append({"workload": f"{cname} {variant} N={Nc} S={Sc}", "error": str(e)[:200]})
    # the headline scene in mixed precision
    try:
        cfg4 = dict(wl.CONFIGS["C4"]); N4 = args.agents
        kw4 = wl.solver_kwargs(cfg4, N4)
        po4, pf4 = wl.make_scenes(cfg4, 1, N4, wl.SEED0 + 4)
        d4m = mp.Dmpc("bound", device=local_rank, precision="mixed", **kw4)
        lp = ClosedLoop(torch, d4m, dev, po4, pf4)
        el = timed(torch, lp.cycle_step, CYCLE * 2, CYCLE)
        add({"workload": f"the headline loop (C4, {N4} agents) in mixed precision (DMPC_PREC_MIXED: fp32 table, scan and rows; fp64 QP)", "timed": "wall clock, device-resident loop",
             "wall_ms_per_mpc_step": el / (CYCLE * 2) * 1e3, "value": N4 * CYCLE * 2 / el, "unit": "solves/s"})
        del d4m, lp
    except Exception as e:   # noqa: BLE001
        sec.append({"workload": "headline loop in mixed precision", "error": str(e)[:200]})
    # one process, every visible GPU (dmpc_create(.., DMPC_DEVICE_ALL, ..): threads + peer copies inside the library) -- the path a MEX / C++
    # caller gets; only when this process sees more than one GPU
    if torch.cuda.device_count() > 1:
        try:
            dg = mp.Dmpc("bound", device=mp.Dmpc.DEVICE_ALL, **kwT)
            dg.transition(poT[:8], pfT[:8], 10, cfgT["error_tol"], histories=False)
            tt = time.perf_counter(); rg = dg.transition(poT, pfT, cfgT["K_T"], cfgT["error_tol"], histories=False); dtg = time.perf_counter() - tt
            add({"workload": f"512 whole transitions, 100 agents, solveSoftDMPCbound, ONE process on {dg.n_devices} GPUs (DMPC_DEVICE_ALL: agents of "
                             "every scene sharded over the GPUs, peer copies between MPC steps)", "n_gpus": dg.n_devices, "wall_ms": dtg * 1e3,
                 "completed": int(((rg["scene_status"] & 256) != 0).sum()), "value": float(((rg["K_T_used"] - 1) * 100).sum() / dtg), "unit": "solves/s"})
            del dg
        except Exception as e:   # noqa: BLE001
            sec.append({"workload": "one process on all GPUs (DMPC_DEVICE_ALL)", "error": str(e)[:200]})
    return sec


if __name__ == "__main__":
    main()
