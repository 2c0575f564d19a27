timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -2
for s in 64 512 1024; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --scenes $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($s, round(d['value']/1e6,2), round(d['ms_per_step'],3), d['roofline']['kernel_ms_avg'])"; done
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/pb -o b -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > /tmp/pb.log 2>&1
python $REPO/tools/pmc_summary.py /tmp/pb | grep -E "solve|scan"
