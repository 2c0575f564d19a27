"""development aid: kernel time vs batch size / variant on the recorded scenes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import multiagent_planning_amd as mp
from helpers import load_golden, step14_inputs

dev = torch.device("cuda", 0)
def run(name, variant, S, reps=5):
    g, kw = load_golden(name)
    l, xp, xv, xa, pf = step14_inputs(g)
    N = l.shape[0]
    d = mp.Dmpc(variant, **kw)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (S,) + a.shape))).to(dev)
    rows = t(l); lT = torch.empty((1, S, 45, N), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    d.table_from_rows_device(S, 1, N, rows.data_ptr(), lT.data_ptr(), st)
    x_p, x_v, x_a, p_f = t(xp), t(xv), t(xa), t(pf)
    po = torch.empty((S, N, 45), dtype=torch.float64, device=dev); vo = torch.empty_like(po); ao = torch.empty_like(po)
    status = torch.zeros((S, N), dtype=torch.int32, device=dev); info = torch.zeros((S, N, 8), dtype=torch.int32, device=dev)
    def step():
        d.step_device(S, 1, N, 0, lT.data_ptr(), x_p.data_ptr(), x_v.data_ptr(), x_a.data_ptr(), p_f.data_ptr(), po.data_ptr(), vo.data_ptr(), ao.data_ptr(), 0, status.data_ptr(), info.data_ptr(), st)
    step(); torch.cuda.synchronize()
    d.profile(True)
    for _ in range(reps): step()
    torch.cuda.synchronize()
    ms, n = d.profile_read()
    inf = info.cpu().numpy()
    print(f"{name:22s} {variant:9s} S={S:4d} agents={S*N:6d} kernel={ms*1e3:9.1f} us  per-agent={ms*1e6/(S*N):8.1f} ns  solves/s={S*N/(ms*1e-3):.3e} iters mean={inf[...,4].mean():.1f} max={inf[...,4].max()}")

if __name__ == "__main__":
    for variant in sys.argv[1:] or ["bound", "hard"]:
        for S in (1, 4, 16, 64):
            run("failure_rate2_bound", variant, S)
