# development: single 64-slot tier against the 32/64 two-tier launch for the slack variants (bench secondaries)
for env in "DMPC_DEBUG_OPTIONS=tier1_qcap=48" "DMPC_DEBUG_OPTIONS=tier1_qcap=32"; do
  echo "== $env"
  env $env timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_t.json
  python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/bench_t.json") if l.startswith("{")][-1])
print("headline", round(j["value"]/1e6,2))
for s in j["secondary"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k in ("value","ms_per_step","mean_iters","max_iters","max_tries","us_per_mpc_step","wall_ms","completed")}, s["workload"][:50])
PY
done
