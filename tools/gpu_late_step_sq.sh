#!/bin/bash
# development: instruction mix of the scan kernel on a LATE step of the bound replay (MPC step 70: nearly every agent is finished by the scan)
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_late; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/sq -o sq -- python $REPO/tools/gpu_bound_ab.py bound f64 ${1:-70} > $OUT/sq.log 2>&1
cd $REPO
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/sq_late/sq/sq_counter_collection.csv")))
# the last dispatches of the scan kernel are the replayed step
scan=[r for r in rows if "scan_kernel" in r["Kernel_Name"]]
last=max(int(r["Dispatch_Id"]) for r in scan)
sel=[r for r in scan if int(r["Dispatch_Id"])==last]
print({r["Counter_Name"].replace("SQ_INSTS_",""): round(float(r["Counter_Value"])/51200,1) for r in sel})
PY
