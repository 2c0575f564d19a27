#!/bin/bash
# development (through gpurun): the text evidence of the second half of round 5 under gpurun_out/r05b/ (copied to profiles/r05_experiments/)
O=gpurun_out/r05b; mkdir -p $O
bash tools/gpu_ifetch_probe.sh r05b_ifetch > $O/ifetch_latency_counters.txt 2>&1
bash tools/gpu_itercap_sweep.sh > $O/itercap_sweep.txt 2>&1
bash tools/gpu_itercap_counters.sh > $O/itercap_instruction_counts.txt 2>&1
python tools/with_trace_lib.py tools/gpu_fixed_cost.py 2>&1 | grep -v amdgpu.ids > $O/fixed_cost_wave_ends_list_scheduling.txt
python tools/with_trace_lib.py tools/gpu_typical_pivots.py 4 12 2>&1 | grep -v amdgpu.ids > $O/typical_pivot_sequences.txt
python tools/with_trace_lib.py tools/gpu_c4_crash_stats.py 4 2>&1 | grep -v amdgpu.ids > $O/crash_stats.txt
bash tools/gpu_opts_ab.sh - queue_chunk=2 no_persist=1 no_split_t=1 order_hint=1 order_hint=2 > $O/options_ab.txt 2>&1
for g in 1 2 4 8; do echo "G=$g $(python bench.py --no-cpu-baseline --no-secondary --emulate-gpus $g 2>/dev/null | tail -1 | python tools/bench_brief.py)"; done > $O/emulated_strong_scaling.txt 2>&1
tail -3 $O/*.txt | cut -c1-300
