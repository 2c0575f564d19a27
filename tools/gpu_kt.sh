#!/bin/bash
# development: rocprofv3 --kernel-trace --stats of a command, per-kernel averages printed.  usage: bash tools/gpu_kt.sh <label> <command...>
L=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$L; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o kt -- "$@" > "$OUT/run.log" 2>&1
cd "$REPO"; find "$OUT" -name "*.db" -delete
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/kt_kernel_stats.csv",recursive=True)
if not f: print("no stats; see $OUT/run.log"); raise SystemExit
for r in csv.DictReader(open(f[0])):
    if float(r["Percentage"]) > 0.15: print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f} %')
PY
tail -2 "$OUT/run.log" | cut -c1-250
