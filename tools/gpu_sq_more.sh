set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_more; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/a -o a -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/c -o c -- $BENCH > $OUT/c.log 2>&1
cd $REPO; find $OUT -name "*.db" -delete
for p in a b c; do python tools/pmc_summary.py $OUT/$p | grep -E "solve|scan"; done
