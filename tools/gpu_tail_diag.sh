mkdir -p gpurun_out/r3_diag
for t in gpu_bound_times gpu_agent_times gpu_wave_times; do
  timeout 600 python tools/with_trace_lib.py tools/$t.py > gpurun_out/r3_diag/$t.log 2>&1
done
tail -n 40 gpurun_out/r3_diag/*.log
