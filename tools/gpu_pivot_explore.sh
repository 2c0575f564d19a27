#!/bin/bash
# development (library built with -DDMPC_PIVOT_EXPLORE as multiagent_planning_amd/libdmpc_hip_explore.so): extra multipliers on the pivot weights,
# DMPC_DEBUG_OPTIONS=pivot_explore=<fields>: 4-bit fields, multiplier 2^(field - 8), 0 = 1; bits 0-3 rows, 4-7 eps <= 0, 8-11 eps >= slb, 12-15 walls, 16-19 bounds
# usage: gpu_pivot_explore.sh hard|soft value ...
K=$1; shift
for v in "$@"; do
  echo "== pivot_explore=$v"
  if [ "$K" = hard ]; then
    DMPC_DEBUG_OPTIONS=pivot_explore=$v timeout 300 python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_explore.so bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python tools/bench_brief.py hard
  else
    DMPC_DEBUG_OPTIONS=pivot_explore=$v python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_explore.so tools/gpu_bound_ab.py bound f64 12 2>&1 | grep -v amdgpu | tail -1
    DMPC_DEBUG_OPTIONS=pivot_explore=$v STEPS=5 python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_explore.so tools/gpu_c4_hist.py 2>&1 | cut -c1-75 | tail -2
  fi
done
