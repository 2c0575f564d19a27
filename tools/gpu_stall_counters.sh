#!/bin/bash
# Development: stall-oriented SQ counters of the solve kernel on the bench workload (two passes)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/stall
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d "$OUT/p1" -o sq -- $BENCH > "$OUT/p1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC \
   --output-format csv -d "$OUT/p2" -o sq -- $BENCH > "$OUT/p2.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" -delete
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for f in glob.glob("gpurun_out/stall/p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "solve_persist" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
wc = res.get("SQ_WAVE_CYCLES", 1)
print(json.dumps({k: [v, round(v / wc, 4)] for k, v in sorted(res.items())}, indent=1))
PY
