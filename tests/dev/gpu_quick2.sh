for s in 512; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --scenes $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($s, round(d['value']/1e6,2), round(d['ms_per_step'],3), d['roofline']['kernel_ms_avg'], d['roofline']['other_kernels_ms_avg'])"; done
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 2 > /tmp/kt.log 2>&1
head -4 /tmp/kt/kt_kernel_stats.csv
