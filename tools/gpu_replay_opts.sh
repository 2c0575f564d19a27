#!/bin/bash
# development (through gpurun): the C2 replays (hard / bound) under sets of debug options -- bash tools/gpu_replay_opts.sh hard|bound "opt=v,opt=v" ...   ("-" = none)
w=$1; shift
for set in "$@"; do
  o=""; [ "$set" != "-" ] && o=$set
  for rep in 1 2; do echo "[$set] $(DMPC_DEBUG_OPTIONS=$o python tools/replay_workload.py $w --steps 40 --warmup 5 2>/dev/null | tail -1)"; done
done
