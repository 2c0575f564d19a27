"""development aid: distribution of iterations / working-set sizes of the C2 bench workload."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "hard"
cfg = dict(wl.CONFIGS["C2"], variant=variant)
S, N = 64, 100
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc(variant, **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, S, N, 12, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
inf, st = out["info"].reshape(-1, 8), out["status"].reshape(-1)
for name, m in (("solved", st == 1), ("infeasible", (st & 8) != 0)):
    it, mq = inf[m, 4], inf[m, 7]
    print(name, m.sum(), "iters mean %.1f p50 %d p90 %d p99 %d max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()),
          "| maxq mean %.1f p90 %d p99 %d max %d  >32: %.3f  >24: %.3f" % (mq.mean(), np.percentile(mq, 90), np.percentile(mq, 99), mq.max(), (mq > 32).mean(), (mq > 24).mean()))
print("total iterations", inf[:, 4].sum(), "share of infeasible agents %.2f" % (inf[(st & 8) != 0, 4].sum() / inf[:, 4].sum()))
