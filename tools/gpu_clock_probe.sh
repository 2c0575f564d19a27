#!/bin/bash
# development (through gpurun): the engine clock while the headline loop runs
python bench.py --no-cpu-baseline --no-secondary --steps 9000 --warmup 9 > /tmp/b.log 2>&1 &
BP=$!
sleep 12
for i in 1 2 3 4; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4
  rocm-smi --showpower --showuse 2>/dev/null | grep -i "power\|busy" | head -3
  cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -4
  sleep 1
done
wait $BP
tail -1 /tmp/b.log | python tools/bench_brief.py
rocm-smi --showperflevel 2>/dev/null | grep -i perf
