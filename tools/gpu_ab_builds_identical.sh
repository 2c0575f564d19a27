#!/bin/bash
# development (through gpurun): are two builds of the library identical in what they return?  Teacher-forced closed loops of the standard workloads
# (tools/gpu_opt_probe.py --dump) with the product build and with multiagent_planning_amd/libdmpc_hip_<name>.so, compared bit for bit (tools/npz_equal.py).
#   usage: bash tools/gpu_ab_builds_identical.sh name
n=$1; O=gpurun_out/ab_$n; mkdir -p $O
for w in "C4 6" "C2b 12" "bound2 12" "cpp 12" "all3 12" "C3 4" "C5 4"; do
  set -- $w
  python tools/gpu_opt_probe.py $1 $2 "" --dump=$O/new_$1.npz 2>/dev/null | tail -1 | cut -c1-200
  python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_$n.so tools/gpu_opt_probe.py $1 $2 "" --dump=$O/old_$1.npz 2>/dev/null | tail -1 | cut -c1-200
  echo "$1: $(python tools/npz_equal.py $O/old_$1.npz $O/new_$1.npz | tail -1)"
done
rm -f $O/*.npz
