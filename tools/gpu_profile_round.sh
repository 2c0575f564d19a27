#!/bin/bash
# Collects the evidence under profiles/ on the GPU box (run through gpurun from the repository root):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload,
#   2. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (HBM-side traffic per launch),
#   3. two SQ counter passes (instruction mix; wave / active / wait cycles),
# then tools/profile_summary.py condenses them into summary.json (per-launch traffic, per-solve instruction mix, issue fractions).
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
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/sq" -o sq -- $BENCH --steps 5 --warmup 1 > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/sq2" -o sq2 -- $BENCH --steps 5 --warmup 1 > "$OUT/sq2.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" -delete   # keep the csv summaries only (size)
python3 tools/profile_summary.py "$OUT" 51200 > "$OUT/summary.json"
tail -1 "$OUT/kt.log" | cut -c1-300
head -c 1500 "$OUT/summary.json"
