"""Transition drivers: the MPC outer loop around the batched step (a11).

  run_transition      -- the loop of dmpc/matlab/dmpc_soft_bound.m:115-148 / test/failure_rate.m:99-127
                         for S scenes on ONE device (thin wrapper of dmpc_transition in the C ABI).
  ShardedStepper      -- agents of every scene sharded across the ranks of a torch.distributed
                         group exactly like the reference's thread clusters (contiguous ranges,
                         dmpc/cpp/dmpc.cpp:1600-1625); per MPC step each rank solves its chunk against the
                         full table and ONE all-gather publishes the new predictions
                         (`l = new_l`, dmpc_soft_bound.m:146; `prev_obs = obs`, dmpc.cpp:1681).

The table layout lT[G][S][3K][C] is rank-major, so the all-gather output IS the next table: no
repacking, no other collective on the data path.  The goal / abort test is a tiny all-reduce(max).
"""
import numpy as np

from ._lib import ST_REACHED, ST_SOLVED

K3 = 45


def partition(N, G):
    """Contiguous clusters: N//G each, the first N % G get one more (dmpc.cpp:1600-1625).
    Returns list of (lo, hi)."""
    base, rem = divmod(N, G)
    out, lo = [], 0
    for r in range(G):
        cnt = base + (1 if r < rem else 0)
        out.append((lo, lo + cnt))
        lo += cnt
    return out


def agent_failed(status):
    """The abort rule of a trial, shared with dmpc_transition / dmpc_transition_sharded of the C ABI (`stbits & ~DMPC_ST_SOLVED`):
    an agent fails the scene unless its status is exactly SOLVED -- infeasible, collided, capacity, and also a SOLVED agent
    whose first step leaves the workspace (SOLVED|OUTBOUND) or that noticed a collision (SOLVED|COLL, the cpp flavour)."""
    return status != ST_SOLVED


def rows_to_chunked(rows, G):
    """[S,N,45] rows -> lT[G,S,45,Cmax], the rank-major table of the C ABI: contiguous clusters as partition(), Cmax =
    ceil(N/G); the last column of the short clusters' chunks is zero padding that nothing reads."""
    S, N, _ = rows.shape
    parts = partition(N, G)
    cmax = parts[0][1] - parts[0][0]
    lT = np.zeros((G, S, K3, cmax), dtype=rows.dtype)
    for g, (lo, hi) in enumerate(parts):
        lT[g, :, :, :hi - lo] = rows[:, lo:hi].transpose(0, 2, 1)
    return lT


def chunked_to_rows(lT, N=None):
    """inverse of rows_to_chunked; N = number of agents (default G*Cmax: equal clusters)"""
    G, S, _, cmax = lT.shape
    N = G * cmax if N is None else N
    return np.ascontiguousarray(np.concatenate([lT[g, :, :, :hi - lo].transpose(0, 2, 1) for g, (lo, hi) in enumerate(partition(N, G))], axis=1))


def run_transition(dmpc, po, pf, K_T_max, error_tol=0.01):
    """Whole transitions of S scenes on one device. Returns dict(pk,vk,ak,K_T_used,scene_status)."""
    return dmpc.transition(po, pf, K_T_max, error_tol)


def run_trial(dmpc, po, pf, K_T_max, error_tol=0.01, vmax=2.0, amax=1.0, Ts=0.01, histories=True):
    """One trial of the reference's test scripts for S scenes (test/failure_rate.m:99-197): the transition loop,
    then -- for scenes that stayed feasible and reached their goals -- the post-checks, which read the histories
    the transition left on the device.  `success` is failure_rate.m:196
    (`feasible && ~failed_goal && ~violation`); t/totdist/traj_time are NaN for failed scenes (:197-201)."""
    tr = dmpc.transition(po, pf, K_T_max, error_tol, histories=histories)   # histories=False: outcomes only, nothing big comes back
    S = tr["K_T_used"].shape[0]
    st = tr["scene_status"]
    feasible = (st & ~ST_REACHED) == ST_SOLVED                 # no agent failed before the scene stopped
    reached = feasible & ((st & ST_REACHED) != 0)              # ReachedGoal.m, evaluated on the device every step
    out = dict(tr, feasible=feasible, failed_goal=feasible & ~reached, violation=np.zeros(S, dtype=np.int32),
               totdist=np.full(S, np.nan), traj_time=np.full(S, np.nan), r_factor=np.full(S, np.nan))
    if reached.any():
        pc = dmpc.postcheck(tr["K_T_used"], pf, KT_alloc=K_T_max, vmax=vmax, amax=amax, Ts=Ts, mask=reached)
        for k in ("violation", "totdist", "traj_time", "r_factor"):
            out[k] = pc[k]
    out["success"] = feasible & reached & (out["violation"] == 0)
    return out


class ShardedStepper:
    """One rank's view of the sharded MPC step.

    local_step(lT_full, x_p, x_v, x_a, pf, g_local) -> dict(p, v, a, status) for this rank's C agents of
    every scene; arrays are numpy on the CPU path used by the gloo tests and torch CUDA tensors on the
    GPU path (see GpuLocalStep).  all_gather is torch.distributed's; with world size 1 it is skipped.
    """

    def __init__(self, local_step, rank, world, group=None):
        self.local_step, self.rank, self.world, self.group = local_step, rank, world, group

    def step(self, lT_full, x_p, x_v, x_a, pf):
        import torch
        import torch.distributed as dist
        out = self.local_step(lT_full, x_p, x_v, x_a, pf, self.rank)
        ok = (out["status"] & 1) == 1
        own = lT_full[self.rank]                                        # [S,45,C]
        newp = out["p"].transpose(-1, -2) if isinstance(out["p"], torch.Tensor) else np.swapaxes(out["p"], -1, -2)
        cnt = newp.shape[-1]                                            # this rank's cluster (<= Cmax columns of its chunk)
        if isinstance(newp, torch.Tensor):
            chunk = own.clone()
            chunk[..., :cnt] = torch.where(ok[:, None, :], newp, own[..., :cnt])
            chunk = chunk.contiguous()
        else:
            ch = np.array(own, copy=True)
            ch[..., :cnt] = np.where(ok[:, None, :], newp, own[..., :cnt])
            chunk = torch.from_numpy(np.ascontiguousarray(ch))
        if self.world > 1:
            nxt = torch.empty((self.world,) + tuple(chunk.shape), dtype=chunk.dtype, device=chunk.device)
            dist.all_gather_into_tensor(nxt.view(-1), chunk.view(-1), group=self.group)   # rank-major == table layout
            flag = torch.tensor([int(agent_failed(out["status"]).any())], dtype=torch.int32, device=chunk.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)   # abort test (failure_rate.m:112-123)
            failed = bool(flag.item())
        else:
            nxt, failed = chunk[None], bool(agent_failed(out["status"]).any())
        if not isinstance(lT_full, torch.Tensor):
            nxt = nxt.numpy()
        return nxt, out, failed


def run_transition_sharded(stepper, lT, x_p, x_v, x_a, pf, K_T_max, error_tol=0.01):
    """The MPC loop of dmpc_soft_bound.m:115-148 / failure_rate.m:99-127 for THIS rank's agents of S scenes, with the
    table exchanged by ShardedStepper.step every step.  A scene stops when one of its agents fails on any rank (the
    reference `break`s the trial) or when every agent of every rank is within error_tol of its goal (ReachedGoal.m):
    both tests are one all-reduce(MAX) of a [2,S] tensor per step.  Inputs are this rank's slices ([S,C,3]) and the full
    chunked table; numpy (gloo) or torch CUDA tensors (RCCL).
    Returns dict(pk [S,C,K_T_max,3] own histories, K_T_used [S], reached [S], failed [S], lT)."""
    import torch
    import torch.distributed as dist
    tens = isinstance(x_p, torch.Tensor)
    xp = torch.as_tensor(x_p).clone(); xv = torch.as_tensor(x_v).clone(); xa = torch.as_tensor(x_a).clone()
    goal = torch.as_tensor(pf)
    S, C = xp.shape[0], xp.shape[1]
    pk = torch.zeros((S, C, K_T_max, 3), dtype=xp.dtype, device=xp.device)
    pk[:, :, 0] = xp
    used = torch.full((S,), K_T_max, dtype=torch.int64)
    reached = torch.zeros(S, dtype=torch.bool); failed = torch.zeros(S, dtype=torch.bool)
    done = torch.zeros(S, dtype=torch.bool)
    for k in range(1, K_T_max):
        lT, out, _ = stepper.step(lT, xp if tens else xp.numpy(), xv if tens else xv.numpy(), xa if tens else xa.numpy(),
                                  goal if tens else goal.numpy())
        st = torch.as_tensor(out["status"]); ok = (st & 1) == 1
        p1, v1, a1 = (torch.as_tensor(out[q])[..., :3] for q in ("p", "v", "a"))
        xp = torch.where(ok[..., None], p1, xp); xv = torch.where(ok[..., None], v1, xv); xa = torch.where(ok[..., None], a1, xa)
        pk[:, :, k] = xp
        flags = torch.stack([agent_failed(st).any(dim=1).to(torch.float64),                        # some agent failed
                             (xp - goal).norm(dim=-1).max(dim=1).values.to(torch.float64)])        # farthest agent from its goal
        flags = flags.to(xp.device)
        if stepper.world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=stepper.group)
        flags = flags.cpu()
        fail_now, reach_now = flags[0] > 0, flags[1] < error_tol
        newly = ~done & (fail_now | reach_now)
        used[newly] = k + 1
        failed |= ~done & fail_now
        reached |= ~done & ~fail_now & reach_now
        done |= newly
        if bool(done.all()):
            break
    conv = (lambda t: t) if tens else (lambda t: t.numpy())
    return dict(pk=conv(pk), K_T_used=used.numpy(), reached=reached.numpy(), failed=failed.numpy(), lT=lT)


class GpuLocalStep:
    """local_step for ShardedStepper on a GPU: torch CUDA tensors in, HIP kernel through the C ABI."""

    def __init__(self, dmpc, S, G, C, device):
        import torch
        self.d, self.S, self.G, self.C = dmpc, S, G, C
        f = dict(dtype=torch.float64, device=device)
        self.p = torch.empty((S, C, K3), **f)
        self.v, self.a = torch.empty_like(self.p), torch.empty_like(self.p)
        self.status = torch.zeros((S, C), dtype=torch.int32, device=device)
        self.info = torch.zeros((S, C, 8), dtype=torch.int32, device=device)

    def __call__(self, lT_full, x_p, x_v, x_a, pf, g_local):
        import torch
        # dmpc_step_device knows equal clusters only (no short_from): a padded table of unequal clusters would be read with the padding
        # columns as real neighbours at the origin -- refuse instead of computing on a wrong layout (unequal clusters: dmpc_step_sharded_device)
        if tuple(lT_full.shape) != (self.G, self.S, K3, self.C) or tuple(x_p.shape[:2]) != (self.S, self.C):
            raise ValueError(f"GpuLocalStep: table {tuple(lT_full.shape)} / state {tuple(x_p.shape)} do not match G={self.G}, S={self.S}, C={self.C} "
                             "(equal clusters N = G*C only; unequal clusters go through Dmpc.step_sharded_device)")
        st = torch.cuda.current_stream().cuda_stream
        self.d.step_device(self.S, self.G, self.C, g_local, lT_full.data_ptr(), x_p.data_ptr(), x_v.data_ptr(), x_a.data_ptr(),
                           pf.data_ptr(), self.p.data_ptr(), self.v.data_ptr(), self.a.data_ptr(), 0, self.status.data_ptr(),
                           self.info.data_ptr(), st)
        return dict(p=self.p, v=self.v, a=self.a, status=self.status, info=self.info)
