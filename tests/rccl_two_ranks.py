"""Helper of tests/test_gpu_multigpu.py::test_two_real_ranks_over_rccl (run under torch.distributed.run with 2 processes, one GPU each):
the in-library RCCL path with MORE THAN ONE real rank -- dmpc_transition_sharded / dmpc_step_sharded_device / the history gather against
the single-GPU dmpc_transition / dmpc_step_batch of the same scenes, bit for bit; rank 0 prints OK."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp          # noqa: E402
from multiagent_planning_amd import workload as wl, _lib   # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
cfg = wl.CONFIGS["C4"]
N, S, KT = 21, 6, 90                           # 21 agents on 2 ranks: unequal clusters
kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 71)
idt = torch.zeros(128, dtype=torch.uint8, device=dev)
if rank == 0:
    idt.copy_(torch.frombuffer(bytearray(mp.Dmpc.comm_unique_id()), dtype=torch.uint8))
dist.broadcast(idt, src=0)
ok = True
for precision in ("f64", "mixed"):
    ref_d = mp.Dmpc("bound", device=local, precision=precision, **kw)
    ref = ref_d.transition(po, pf, KT, cfg["error_tol"])
    d = mp.Dmpc("bound", device=local, precision=precision, **kw)
    if precision == "mixed":
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(mp.Dmpc.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
    d.comm_init(bytes(idt.cpu().numpy().tobytes()), world, rank)
    out = d.transition_sharded(po, pf, KT, cfg["error_tol"], gather=True)
    lo, cnt = out["lo"], out["count"]
    ok &= bool(np.array_equal(out["K_T_used"], ref["K_T_used"]) and np.array_equal(out["scene_status"], ref["scene_status"]))
    for k in ("pk", "vk", "ak"):
        ok &= bool(np.array_equal(out[k], ref[k][:, lo:lo + cnt]))
    pc = d.postcheck(out["K_T_used"], pf, KT_alloc=KT)             # the gathered scene-wide histories, on every rank
    pc_ref = ref_d.postcheck(ref["K_T_used"], pf, KT_alloc=KT)
    for k in pc:
        ok &= bool(np.array_equal(pc[k], pc_ref[k], equal_nan=True))
    d.comm_destroy()
flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
if rank == 0:
    print("RCCL_TWO_RANKS_OK" if int(flag.item()) == 1 else "RCCL_TWO_RANKS_MISMATCH", flush=True)
sys.exit(0 if int(flag.item()) == 1 else 1)
