#!/bin/bash
# development: the round's standard quick check on the GPU box: the whole -m gpu suite (pytest-xdist off: one GPU) and a short bench
# usage: gpurun -- 'bash tools/gpu_quick.sh <label> [notests]'
L=${1:-q}; mkdir -p gpurun_out/$L
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$L/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$L/tests.log
  tail -n 6 gpurun_out/$L/tests.log
fi
timeout 900 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/$L/bench.json 2> gpurun_out/$L/bench.err
python tools/bench_brief.py $L < gpurun_out/$L/bench.json
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$L/bench.json") if l.startswith("{")][-1])
for s in d.get("secondary") or []:
    print("  ", s["workload"][:90], "|", {k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k in ("value","ms_per_step","us_per_mpc_step","wall_ms","ms_per_mpc_step","max_iters","first_step_ms")})
PY
