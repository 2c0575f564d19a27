mkdir -p gpurun_out/r3g
timeout 1200 python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_precision.py tests/test_mex_gateway.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r3g/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3g/tests.log
tail -n 30 gpurun_out/r3g/tests.log
