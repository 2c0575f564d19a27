#!/bin/bash
# development: kernel trace of ONE 100-agent scene in closed loop (where do the microseconds of a step go)
REPO=$(pwd); OUT=$REPO/gpurun_out/single_trace; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ss -- python $REPO/tools/gpu_single_scene.py > "$OUT/log.txt" 2>&1
cd "$REPO"
tail -3 "$OUT/log.txt"
head -12 "$OUT/ss_kernel_stats.csv" | cut -c1-200
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/single_trace/ss_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# gaps between consecutive kernels
gaps=collections.defaultdict(list)
for a,b in zip(rows,rows[1:]):
    g=int(b["Start_Timestamp"])-int(a["End_Timestamp"])
    if g<200000: gaps[(a["Kernel_Name"][:40],b["Kernel_Name"][:40])].append(g)
for k,v in sorted(gaps.items(),key=lambda kv:-len(kv[1]))[:12]:
    v.sort(); print(k,len(v),"median gap ns",v[len(v)//2],"mean",sum(v)//len(v))
PY
