# development: crash start / ladder warm start on and off -- tests, C4 N = 10^4 statistics, headline, one-scene latency
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
for env in "X=1" "DMPC_DEBUG_OPTIONS=crash_min=0"; do
  echo "== $env"
  env $env STEPS=5 timeout 300 python tools/gpu_c4_hist.py 2>&1 | sed 's/| with rows.*| tries/| tries/' | cut -c1-330
  env $env timeout 300 python tools/gpu_single_scene.py 2>&1 | tail -2
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_sec.json; python tools/bench_brief.py headline < gpurun_out/bench_sec.json
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/bench_sec.json") if l.startswith("{")][-1])
for s in j["secondary"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k in ("value","ms_per_step","mean_iters","max_iters","max_tries","us_per_mpc_step","wall_ms","completed")}, s["workload"][:60])
PY
