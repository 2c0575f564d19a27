#!/bin/bash
# development (through gpurun): instructions per solve of the headline launch (one SQ pass) -- bash tools/gpu_inst_counts.sh [lib.so]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/inst_counts; rm -rf "$OUT"; mkdir -p "$OUT"
LIB=${1:-}
cd /tmp && export TMPDIR=/tmp
if [ -n "$LIB" ]; then CMD="python $REPO/tools/with_lib.py $REPO/$LIB $REPO/bench.py"; else CMD="python $REPO/bench.py"; fi
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/e" -o e -- $CMD --no-cpu-baseline --no-secondary --steps 9 --warmup 0 > "$OUT/e.log" 2>&1
cd "$REPO"; find "$OUT" -name "*.db" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "solve_persist_kernel<true, 56" in r["Kernel_Name"] or "rsolve_persist" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    d = {c: round(v / n[c] / 10000, 1) for c, v in acc.items()}
    print(d, "total", round(sum(v for k, v in d.items() if k != "SQ_WAVES")))
PY
