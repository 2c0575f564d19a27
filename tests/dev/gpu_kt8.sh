REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -o kt -- python $REPO/bench.py --no-cpu-baseline --no-secondary --emulate-gpus ${1:-8} --steps 10 --warmup 2 > /tmp/kt8.log 2>&1
head -8 /tmp/kt8/kt_kernel_stats.csv | cut -c1-150
