"""development (round 4): what ONE rank of the C4 strong-scaling run has to do per MPC step -- the 10^4-agent scene as G chunks, chunk 0 solved on this GPU
(table of all chunks resident, as after the all-gather): device time per step for G = 1, 2, 4, 8.  No exchange is timed: the bound on the scaling curve
that the solver alone sets (a rank's launch cannot end before its longest agent does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg, N, S = dict(wl.CONFIGS["C4"]), 10000, 1
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
d0 = mp.Dmpc("bound", **kw)
l, _, _ = d0.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
states = []
for k in range(6):
    states.append((l, xp, xv, xa))
    o = d0.step_batch(l, xp, xv, xa, pf)
    ok = (o["status"] == 1)[..., None]
    l = np.where(ok, o["p"], l); xp = np.where(ok, o["p"][..., :3], xp); xv = np.where(ok, o["v"][..., :3], xv); xa = np.where(ok, o["a"][..., :3], xa)
states = states[1:]
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev, torch.float64)
for G in (1, 2, 4, 8):
    C = N // G
    d = mp.Dmpc("bound", **kw)
    bufs = []
    for (l_, xp_, xv_, xa_) in states:
        rows = t(l_); lT = torch.empty((G, S, 45, C), dtype=torch.float64, device=dev)
        d.table_from_rows_device(S, G, C, rows.data_ptr(), lT.data_ptr(), stream)
        bufs.append((lT, t(xp_[:, :C]), t(xv_[:, :C]), t(xa_[:, :C])))
    pft = t(pf[:, :C])
    p = torch.empty((S, C, 45), dtype=torch.float64, device=dev); v, a = torch.empty_like(p), torch.empty_like(p)
    st = torch.zeros((S, C), dtype=torch.int32, device=dev); inf = torch.zeros((S, C, 8), dtype=torch.int32, device=dev)
    def step(i):
        lT, a1, a2, a3 = bufs[i % len(bufs)]
        d.step_device(S, G, C, 0, lT.data_ptr(), a1.data_ptr(), a2.data_ptr(), a3.data_ptr(), pft.data_ptr(), p.data_ptr(), v.data_ptr(), a.data_ptr(), 0, st.data_ptr(), inf.data_ptr(), stream)
    for i in range(5): step(i)
    torch.cuda.synchronize(); d.profile(True)
    for i in range(20): step(i)
    torch.cuda.synchronize()
    sms, cms, n = d.profile_read2()
    print(f"G = {G}: chunk of {C} agents: solve {sms * 1e3:6.0f} us + scan / lists / order {cms * 1e3:5.0f} us = {(sms + cms) * 1e3:6.0f} us per MPC step; longest agent of the chunk {inf.cpu().numpy()[..., 4].max()} iterations")
