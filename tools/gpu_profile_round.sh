#!/bin/bash
# Collects the evidence under profiles/ on the GPU box (run through gpurun from the repository root):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload,
#   2. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (HBM-side traffic per launch),
#   3. one SQ counter pass (instruction mix, VALU activity).
# Every pass is bounded by `timeout`; counters never share a run with trace domains other than --kernel-trace.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/profile_round
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH --steps 30 --warmup 3 > "$OUT/kt.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH --steps 5 --warmup 1 > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $BENCH --steps 5 --warmup 1 > "$OUT/write.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$OUT/sq" -o sq -- $BENCH --steps 5 --warmup 1 > "$OUT/sq.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" -delete   # keep the csv summaries only (size)
tail -1 "$OUT/kt.log"
find "$OUT" -name "*.csv" | head -20
