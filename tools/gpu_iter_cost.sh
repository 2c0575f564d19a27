#!/bin/bash
# Development: instruction cost of one active-set iteration of the solve kernel.  The SQ instruction counters of the bench
# workload are collected with the iteration cap (development option iter_cap) at 0, 4 and unlimited; the differences divided by the
# differences of the mean iteration count give the per-iteration cost, the cap-0 run the fixed per-agent cost.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/iter_cost
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1"
for cap in 0 4 100000; do
  DMPC_DEBUG_OPTIONS=iter_cap=$cap timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d "$OUT/cap$cap" -o sq -- $BENCH > "$OUT/cap$cap.log" 2>&1
done
cd "$REPO"
find "$OUT" -name "*.db" -delete
python3 - <<'PY'
import csv, json, glob, os, collections
out = {}
for d in sorted(glob.glob("gpurun_out/iter_cost/cap*/")):
    cap = os.path.basename(d.rstrip("/"))
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "solve" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = [l for l in open(f"gpurun_out/iter_cost/{cap}.log") if l.startswith("{")]
    ws = json.loads(line[-1])["workload_stats"] if line else {}
    out[cap] = {"per_solve": {k: sum(v) / len(v) / 51200 for k, v in acc.items()}, "mean_iters": ws.get("mean_iters")}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/iter_cost/summary.json", "w"), indent=1)
PY
