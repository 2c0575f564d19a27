#!/bin/bash
# development: the headline launch against compile-time variants of the library (multiagent_planning_amd/libdmpc_hip_<name>.so)
# usage: gpurun -- 'bash tools/gpu_ab_libs.sh name1 name2 ...'   ("base" = the product build)
for n in "$@"; do
  for rep in 1 2; do
    if [ "$n" = base ]; then timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 40 | python tools/bench_brief.py $n
    else timeout 300 python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_$n.so bench.py --no-cpu-baseline --no-secondary --steps 40 | python tools/bench_brief.py $n; fi
  done
done
