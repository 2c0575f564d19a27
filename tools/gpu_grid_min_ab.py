"""development (round 4): from which scene size on the cell-grid lists beat the all-pairs box test -- scan side (lists + scan + order) of one
replayed first solve, scenes of N agents, S scenes with S N ~ 4 10^5 (the per-rank load of the weak-scaling bench), both list passes.
usage: python tools/gpu_grid_min_ab.py [variant]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
variant = sys.argv[1] if len(sys.argv) > 1 else "hard"
dev = torch.device("cuda", 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev, torch.float64)
stream = torch.cuda.current_stream().cuda_stream
for N, S in ((400, 256), (800, 128), (1600, 64), (3200, 32)):
    cfg = dict(wl.CONFIGS["C2"], variant=variant)
    kw = wl.solver_kwargs(cfg, N)
    po, pf = wl.make_scenes(cfg, 4, N, wl.SEED0 + 7)
    po, pf = np.tile(po, (S // 4, 1, 1)), np.tile(pf, (S // 4, 1, 1))
    out = []
    for grid_min in (1 << 30, 256):
        d = mp.Dmpc(variant, **kw)
        d.debug_option("grid_min", grid_min)
        l, _, _ = d.init_batch(po, pf)
        rows = t(l); lT = torch.empty((1, S, 45, N), dtype=torch.float64, device=dev)
        d.table_from_rows_device(S, 1, N, rows.data_ptr(), lT.data_ptr(), stream)
        xp, z, pft = t(po), t(np.zeros_like(po)), t(pf)
        p = torch.empty((S, N, 45), dtype=torch.float64, device=dev); v, ac = torch.empty_like(p), torch.empty_like(p)
        st = torch.zeros((S, N), dtype=torch.int32, device=dev)
        def step():
            d.step_device(S, 1, N, 0, lT.data_ptr(), xp.data_ptr(), z.data_ptr(), z.data_ptr(), pft.data_ptr(), p.data_ptr(), v.data_ptr(), ac.data_ptr(), 0, st.data_ptr(), 0, stream)
        for _ in range(2): step()
        torch.cuda.synchronize(); d.profile(True)
        for _ in range(6): step()
        torch.cuda.synchronize()
        sms, cms, n = d.profile_read2()
        out.append((cms, sms, p.clone(), st.clone()))
    same = bool(torch.equal(out[0][2], out[1][2])) and bool(torch.equal(out[0][3], out[1][3]))
    print(f"{variant} N={N} x {S} scenes: scan side all-pairs {out[0][0]:.3f} ms, grid {out[1][0]:.3f} ms (solve {out[0][1]:.3f} / {out[1][1]:.3f}); identical results: {same}")
