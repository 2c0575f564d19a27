#!/bin/bash
# development (CPU only): static instruction mix and resources of one kernel of the library (default: the headline's solve kernel)
#   usage: bash tools/isa_stats.sh [mangled-name-substring] [extra hipcc flags...]
K=${1:-25dmpc_solve_persist_kernelILb1ELi56ELi48EdEE}; shift
OUT=/tmp/isa; mkdir -p $OUT
cd "$(dirname "$0")/../multiagent_planning_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" --cuda-device-only -S -o $OUT/dmpc.s dmpc_api.hip 2>/dev/null || exit 1
awk -v k="$K" '$0 ~ "^_ZN4dmpc"k".*:" {on=1} on {print} on && /^\.Lfunc_end/ {exit}' $OUT/dmpc.s > $OUT/k.s
awk -v k="$K" '$0 ~ "^_ZN4dmpc"k".*:" {on=1} on && /; (NumVgprs|ScratchSize|Occupancy|codeLenInByte|SGPRBlocks|NumSgprs)/ {print} on && /; Occupancy/ {exit}' $OUT/dmpc.s | tr '\n' ' '; echo
printf "instr %d  valu %d  salu %d  lds %d  vmem %d  smem %d  branch %d  waitcnt %d  nop %d  readlane %d  writelane %d\n" \
  $(grep -cE "^\s+[sv]_|^\s+ds_|^\s+global_|^\s+flat_|^\s+buffer_|^\s+scratch_" $OUT/k.s) $(grep -cE "^\s+v_" $OUT/k.s) $(grep -cE "^\s+s_" $OUT/k.s) $(grep -cE "^\s+ds_" $OUT/k.s) \
  $(grep -cE "^\s+(global|flat|buffer|scratch)_" $OUT/k.s) $(grep -cE "^\s+s_(load|buffer_load)" $OUT/k.s) $(grep -cE "^\s+s_c?branch" $OUT/k.s) $(grep -c "s_waitcnt" $OUT/k.s) $(grep -c "s_nop" $OUT/k.s) $(grep -c v_readlane $OUT/k.s) $(grep -c v_writelane $OUT/k.s)
