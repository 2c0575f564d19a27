"""development aid: achieved HBM bandwidth of the dense pairwise row builder (dmpc_add_coll_constr_device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import multiagent_planning_amd as mp

N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 50
colmajor = len(sys.argv) > 3 and sys.argv[3] == "F"
d = mp.Dmpc("bound")
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
p = torch.from_numpy(rng.uniform(-2, 2, (N, K, 3))).to(dev)
po = p[:, 0].contiguous()
A = torch.from_numpy(np.kron(np.eye(N), mp.model_matrices(0.2, K)[0])).to(dev)
ncols, nrows = 3 * K * N, K * N * (N - 1) // 2
Ain = torch.empty((nrows, ncols) if not colmajor else (ncols, nrows), dtype=torch.float64, device=dev)
b = torch.empty(nrows, dtype=torch.float64, device=dev)
o_rs, o_cs = (ncols, 1) if not colmajor else (1, nrows)
a_rs, a_cs = (ncols, 1) if not colmajor else (1, 3 * K * N)      # symmetric block-diagonal A: either view is the same matrix
st = torch.cuda.current_stream().cuda_stream
def run():
    d._chk(d._L.dmpc_add_coll_constr_device(d._ctx, K, N, p.data_ptr(), po.data_ptr(), 0.5, 2.0, A.data_ptr(), a_rs, a_cs, ncols,
                                             Ain.data_ptr(), o_rs, o_cs, b.data_ptr(), st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
out_b = nrows * ncols * 8
e0.record()
for _ in range(reps):
    Ain.fill_(1.0)
e1.record(); torch.cuda.synchronize()
fill_ms = e0.elapsed_time(e1) / reps
print(f"N={N} K={K} {'col' if colmajor else 'row'}-major: rows={nrows} cols={ncols} out={out_b/1e6:.1f} MB A={A.numel()*8/1e6:.1f} MB  "
      f"{ms*1e3:.1f} us  write {out_b/ms/1e6:.0f} GB/s  ({out_b/ms/1e6/8000*100:.1f}% of 8 TB/s); torch fill_ of the same buffer {out_b/fill_ms/1e6:.0f} GB/s")
