import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, collections
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "all3"
cfg, N, S = dict(wl.CONFIGS["C2"], variant=variant), 100, 512
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
st = out["status"].reshape(-1); inf = out["info"].reshape(-1, 8)
print("status histogram", collections.Counter(st.tolist()).most_common(8))
it = inf[:, 4]
for a in np.argsort(it)[::-1][:12]:
    print(f"agent {a}: status {st[a]} viol_k {inf[a,0]} rows {inf[a,1]} tries {inf[a,2]} iters {inf[a,4]} nslack {inf[a,5]} q {inf[a,6]} maxq {inf[a,7]}")
print("iters: sum", it.sum(), "agents >100:", (it > 100).sum(), "their sum", it[it > 100].sum(), "| tries>1:", (inf[:,2] > 1).sum(), "maxq>48:", (inf[:,7] > 48).sum())
