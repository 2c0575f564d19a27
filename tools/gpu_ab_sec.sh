#!/bin/bash
# development: the secondary workloads of the bench (replays, transitions, C4) against compile-time variants of the library
# usage: gpurun -- 'bash tools/gpu_ab_sec.sh name1 name2 ...'   ("base" = the product build)
for n in "$@"; do
  echo "== $n"
  if [ "$n" = base ]; then timeout 600 python bench.py --no-cpu-baseline --steps 20 > /tmp/ab_$n.json 2>/dev/null
  else timeout 600 python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_$n.so bench.py --no-cpu-baseline --steps 20 > /tmp/ab_$n.json 2>/dev/null; fi
  python - <<PY
import json
d=json.loads([l for l in open("/tmp/ab_$n.json") if l.startswith("{")][-1])
print("   headline %.2f M/s" % (d["value"]/1e6))
for s in d.get("secondary") or []:
    print("  ", s["workload"][:70], "|", {k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k in ("value","ms_per_step","us_per_mpc_step","wall_ms","ms_per_mpc_step","max_iters")})
PY
done
