timeout 1500 python tests/dev/gpu_campaign.py 250 1 2>&1 | grep -v amdgpu | tail -8
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_t.json; python tools/bench_brief.py cond < gpurun_out/bench_t.json
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/bench_t.json") if l.startswith("{")][-1])
for s in j["secondary"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k in ("value","ms_per_step","mean_iters","max_iters","us_per_mpc_step","wall_ms","completed","ms_per_mpc_step")}, s["workload"][:40])
PY
