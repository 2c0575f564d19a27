set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/scan_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/a -o a -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH_LEVEL SQ_IFETCH SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU2 SQ_INSTS --output-format csv -d $OUT/c -o c -- $BENCH > $OUT/c.log 2>&1
cd $REPO; find $OUT -name "*.db" -delete
for p in a b c; do python tools/pmc_summary.py $OUT/$p | grep -E "solve|scan"; tail -2 $OUT/$p.log | cut -c1-200; done
