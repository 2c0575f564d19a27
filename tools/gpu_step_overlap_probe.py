"""development aid: would two half-batches of the headline step on two streams beat one full-batch step?"""
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
cfg = wl.CONFIGS["C2"]; N = 100; S = 512
kw = wl.solver_kwargs(cfg, N)
d0 = mp.Dmpc("hard", **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d0, cfg, S, N, 12, wl.SEED0 + 2)
dev = torch.device("cuda", 0)
def setup(sl):
    d = mp.Dmpc("hard", **kw)
    Sg = sl.stop - sl.start
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
    rows = t(l); lT = torch.empty((1, Sg, 45, N), dtype=torch.float64, device=dev)
    d.table_from_rows_device(Sg, 1, N, rows.data_ptr(), lT.data_ptr(), 0)
    st = torch.cuda.Stream(device=dev)
    bufs = dict(lT=lT, x=[t(xp), t(xv), t(xa), t(pf)], p=torch.empty((Sg, N, 45), dtype=torch.float64, device=dev),
                v=torch.empty((Sg, N, 45), dtype=torch.float64, device=dev), a=torch.empty((Sg, N, 45), dtype=torch.float64, device=dev),
                nx=torch.empty((Sg, 45, N), dtype=torch.float64, device=dev), s=torch.zeros((Sg, N), dtype=torch.int32, device=dev),
                i=torch.zeros((Sg, N, 8), dtype=torch.int32, device=dev))
    def step():
        d.step_device(Sg, 1, N, 0, bufs["lT"].data_ptr(), *[x.data_ptr() for x in bufs["x"]], bufs["p"].data_ptr(), bufs["v"].data_ptr(),
                      bufs["a"].data_ptr(), bufs["nx"].data_ptr(), bufs["s"].data_ptr(), bufs["i"].data_ptr(), st.cuda_stream)
    return step
torch.cuda.synchronize()
for G in (1, 2, 4):
    steps = [setup(slice(g * S // G, (g + 1) * S // G)) for g in range(G)]
    for _ in range(3):
        for f in steps: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 30
    for _ in range(reps):
        for f in steps: f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{G} stream(s) x {S//G} scenes: {dt*1e3:.3f} ms per step of {S*N} QPs = {S*N/dt/1e6:.2f} M solves/s")
