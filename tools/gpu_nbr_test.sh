# development: large-scene scan path (neighbour pre-pass + list walk): tests, C4 N = 10^4 steps, 8-rank weak-scaling scene, headline
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -5
STEPS=9 timeout 300 python tools/gpu_c4_hist.py 2>&1 | cut -c1-110
timeout 300 python bench.py --no-cpu-baseline --no-secondary --emulate-gpus 8 | python tools/bench_brief.py emu8
timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py headline
timeout 300 python tools/gpu_configs.py C3 C5 2>&1 | cut -c1-250
