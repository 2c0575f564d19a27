# development: per-kernel times of the C4 N = 10^4 steps
REPO=$(pwd); OUT=$REPO/gpurun_out/c4_trace; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=9 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o c4 -- python $REPO/tools/gpu_c4_hist.py > "$OUT/log.txt" 2>&1
cd $REPO
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/c4_trace/c4_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    n=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if d>20: print(f"{d:9.1f} us  {n[:70]}")
PY
