#!/bin/bash
# development: the kernels of one MPC step of a batched transition (512 scenes x 100 agents, solveSoftDMPCbound, 4 batch parts): names, durations, gaps
REPO=$(pwd); OUT=$REPO/gpurun_out/tr_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tr.py <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg = dict(wl.CONFIGS["C4"]); kw = wl.solver_kwargs(cfg, 100)
d = mp.Dmpc("bound", **kw)
po, pf = wl.make_scenes(cfg, 512, 100, wl.SEED0 + 100)
d.transition(po[:64], pf[:64], 10, cfg["error_tol"], histories=False)
for _ in range(2): r = d.transition(po, pf, cfg["K_T"], cfg["error_tol"], histories=False)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python /tmp/tr.py > $OUT/t.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/tr_trace/t/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t_end=int(rows[-1]["End_Timestamp"])
# the last transition: take the last 40 % of the time span
t0=int(rows[0]["Start_Timestamp"]); span=t_end-t0
sel=[r for r in rows if int(r["Start_Timestamp"])>t_end-0.25*span]
agg=collections.defaultdict(lambda:[0,0.0])
for r in sel:
    n=r["Kernel_Name"].split("(")[0][:60]; agg[n][0]+=1; agg[n][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=(int(sel[-1]["End_Timestamp"])-int(sel[0]["Start_Timestamp"]))/1e3
print(f"window {tot:.0f} us, kernels {len(sel)}, sum of kernel durations {sum(v[1] for v in agg.values()):.0f} us")
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(f"{t:9.0f} us {c:5d} x {t/c:7.1f}  {n}")
# per queue (stream) the sequence of one step
q=sel[len(sel)//2]["Queue_Id"]
seq=[r for r in sel if r["Queue_Id"]==q][:14]
print("one stream:")
for a,b in zip(seq,seq[1:]):
    print(f'  {(int(a["End_Timestamp"])-int(a["Start_Timestamp"]))/1e3:7.1f} us  gap to next {(int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3:6.1f}  {a["Kernel_Name"].split("(")[0][:50]}')
PY
