#!/bin/bash
# Collects the evidence under profiles/ on the GPU box (run through gpurun from the repository root):
#   1. rocprofv3 --kernel-trace --stats of the workload,
#   2. FETCH_SIZE / WRITE_SIZE in separate --pmc passes (HBM-side traffic per launch),
#   3. three SQ counter passes (instruction mix; wave / active / wait cycles; fp64 arithmetic classes + lane utilisation),
# then tools/profile_summary.py condenses them into summary.json (per-launch traffic, per-solve instruction mix, issue fractions,
# fp64 share of the VALU instructions, achieved fp64 FLOP/s).  Every pass is bounded by `timeout`; counters never share a run with trace
# domains other than --kernel-trace.
#   usage: bash tools/gpu_profile_round.sh [label] [workload]     workload: headline (bench.py default since round 5: C4, ONE scene of 10^4 agents,
#          solveSoftDMPCbound, device-resident closed loop over MPC steps 2-10) | c2 (100 agents x 512 scenes of solveHardDMPC: the headline of
#          rounds 1-4) | bound (the C2 scenes with solveSoftDMPCbound, MPC step 12)
set -u
LABEL=${1:-profile_round}; WL=${2:-headline}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$LABEL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case $WL in
  headline|c4) BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary"; SPL=10000; KERN="rsolve_persist_kernel"; WPS=2.0;;
  c2)       BENCH="python $REPO/tools/replay_workload.py hard"; SPL=51200; KERN="solve_persist_kernel<false, 48"; WPS=3;;
  bound)    BENCH="python $REPO/tools/replay_workload.py bound"; SPL=51200; KERN="solve_persist_kernel<true, 48"; WPS=2;;
esac
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH --steps 27 --warmup 9 > "$OUT/kt.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH --steps 9 --warmup 0 > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $BENCH --steps 9 --warmup 0 > "$OUT/write.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/sq" -o sq -- $BENCH --steps 9 --warmup 0 > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/sq2" -o sq2 -- $BENCH --steps 9 --warmup 0 > "$OUT/sq2.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/sq3" -o sq3 -- $BENCH --steps 9 --warmup 0 > "$OUT/sq3.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib" -o calib -- python $REPO/tools/gpu_fetch_calib.py > "$OUT/calib.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" -delete   # keep the csv summaries only (size)
python3 tools/profile_summary.py "$OUT" $SPL "$KERN" $WPS > "$OUT/summary.json"
tail -1 "$OUT/kt.log" | cut -c1-300
python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
print({k: d.get(k) for k in ("solve_kernel", "source_hash", "instructions_per_solve", "fp64", "wave_time", "traffic_calibration")})
print(d.get("kernel_stats"))
PY
