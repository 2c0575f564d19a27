# development: kernel statistics of the whole default bench (headline + secondaries)
REPO=$(pwd); OUT=$REPO/gpurun_out/bound_trace; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o b -- python $REPO/bench.py --no-cpu-baseline > "$OUT/log.txt" 2>&1
cd $REPO; cut -c1-160 "$OUT/b_kernel_stats.csv" | head -24
