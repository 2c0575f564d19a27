timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py vrows
DMPC_NO_VROWS=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py materialized
timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py vrows
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error\|assert" gpurun_out/pytest_gpu.log | tail -8
