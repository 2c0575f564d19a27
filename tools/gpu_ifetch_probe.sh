#!/bin/bash
# development (through gpurun, from the repository root): instruction-fetch and latency counters of the headline launch -- does the 57 KB
# solve kernel (+ 9 KB crash_append) live in the 64 KB instruction cache two CUs share, and what do an LDS / scalar / vector memory
# instruction wait for on average?  Counters in their own passes with --kernel-trace only.
#   usage: bash tools/gpu_ifetch_probe.sh [label]
set -u
LABEL=${1:-ifetch_probe}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$LABEL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 9 --warmup 0"
run() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o $name -- $BENCH > "$OUT/$name.log" 2>&1; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run lat1 InstrFetchLatency
run lat2 LdsLatency
run lat3 SmemLatency
run lat4 VmemLatency
run wt SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_LDS
cd "$REPO"
find "$OUT" -name "*.db" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*counter_collection.csv") + glob.glob(out + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print(f)
    for k in acc:
        if "solve" in k or "scan_kernel" in k or "grid_query" in k:
            print("  ", k, {c: round(v / n[(k, c)], 2) for c, v in acc[k].items()})
PY
