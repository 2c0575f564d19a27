#!/usr/bin/env python3
"""development: the secondary workloads of bench.py as stand-alone replay loops for the profiler passes of tools/gpu_profile_round.sh
(usage: replay_workload.py hard|bound|c4 --steps K --warmup W).
  hard:  C2 (BASELINE configs[1], the headline of rounds 1-4): 100 agents x 512 scenes of solveHardDMPC, replay of the captured first solve
  bound: the same scenes (C2 box, 100 agents x 512 scenes) with the reference's primary variant solveSoftDMPCbound, replay of MPC step 12
  c4:    ONE scene of 10^4 agents, solveSoftDMPCbound, teacher-forced MPC steps 2-10 replayed round-robin"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
ap = argparse.ArgumentParser(); ap.add_argument("what"); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--one-state", type=int, default=-1, help="c4: replay this one state only (development: the order hint with a perfect prediction)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
t = lambda x, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(x)).to(dev, dt)
stream = torch.cuda.current_stream().cuda_stream
if a.what in ("bound", "hard"):
    cfg, N, S = dict(wl.CONFIGS["C2"], variant=a.what), 100, 512
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc(a.what, **kw)
    l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
    states = [(l, xp, xv, xa)]
else:
    cfg, N, S = dict(wl.CONFIGS["C4"]), 10000, 1
    kw = wl.solver_kwargs(cfg, N)
    d = mp.Dmpc("bound", **kw)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
    l, _, _ = d.init_batch(po, pf)
    xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
    states = []
    for k in range(9):
        states.append((l, xp, xv, xa))
        o = d.step_batch(l, xp, xv, xa, pf)
        ok = o["status"] == 1
        l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
        xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
    states = states[1:]          # MPC steps 3-10 (the first solve from the initDMPC table is the outlier)
    if a.one_state >= 0: states = [states[a.one_state]]
bufs = []
for (l_, xp_, xv_, xa_) in states:
    rows = t(l_); lT = torch.empty((1, S, 45, N), dtype=torch.float64, device=dev)
    d.table_from_rows_device(S, 1, N, rows.data_ptr(), lT.data_ptr(), stream)
    bufs.append((lT, t(xp_), t(xv_), t(xa_)))
pft = t(pf)
p = torch.empty((S, N, 45), dtype=torch.float64, device=dev); v, ac = torch.empty_like(p), torch.empty_like(p)
nxt = torch.empty((S, 45, N), dtype=torch.float64, device=dev)
st = torch.zeros((S, N), dtype=torch.int32, device=dev); inf = torch.zeros((S, N, 8), dtype=torch.int32, device=dev)
def step(i):
    lT, a1, a2, a3 = bufs[i % len(bufs)]
    d.step_device(S, 1, N, 0, lT.data_ptr(), a1.data_ptr(), a2.data_ptr(), a3.data_ptr(), pft.data_ptr(), p.data_ptr(), v.data_ptr(), ac.data_ptr(),
                  nxt.data_ptr(), st.data_ptr(), inf.data_ptr(), stream)
for i in range(a.warmup):
    step(i)
torch.cuda.synchronize()
d.profile(True)
t0 = time.perf_counter()
for i in range(a.steps):
    step(i)
torch.cuda.synchronize()
el = time.perf_counter() - t0
sms, cms, n = d.profile_read2()
print(f"{a.what}: {S * N * a.steps / el / 1e6:.2f} M solves/s, {el / a.steps * 1e3:.3f} ms per step (solve {sms:.3f} ms, scan+order {cms:.3f} ms); "
      f"mean iters {inf.cpu().numpy()[..., 4].mean():.2f}")
