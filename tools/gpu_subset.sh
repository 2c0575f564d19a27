#!/bin/bash
# development: a subset of the -m gpu suite on the GPU box:  gpurun -- 'bash tools/gpu_subset.sh "tests/test_gpu_certificates.py tests/test_gpu_paths.py" [label]'
L=${2:-subset}; mkdir -p gpurun_out/$L
timeout 1500 python -m pytest $1 -m gpu -x -q --durations=8 > gpurun_out/$L/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$L/tests.log
tail -n 25 gpurun_out/$L/tests.log
