#!/bin/bash
# development (through gpurun): the headline bench under sets of debug options -- bash tools/gpu_opts_ab.sh "opt=v opt=v" "opt=v" ...   ("-" = none)
for set in "$@"; do
  o=""; [ "$set" != "-" ] && for kv in $set; do o="$o --debug-option $kv"; done
  for rep in 1 2; do echo "[$set] $(python bench.py --no-cpu-baseline --no-secondary --steps 36 $o 2>/dev/null | tail -1 | python tools/bench_brief.py)"; done
done
