# development: three seeds of the randomized parity campaign (all variants), then tests and the bench
timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py acc
for s in 1 2 3; do timeout 1200 python tests/dev/gpu_campaign.py 250 $s 2>&1 | grep -v amdgpu | tail -4; done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|FAILED\|Fatal" gpurun_out/pytest_gpu.log | tail -4
