for env in "X=1" "DMPC_DEBUG_OPTIONS=no_persist=1" "DMPC_DEBUG_OPTIONS=crash_min=0" "DMPC_DEBUG_OPTIONS=tier1_qcap=64"; do
  echo "== $env"
  env $env STEPS=5 timeout 300 python tools/gpu_c4_hist.py 2>&1 | sed 's/| with rows.*| tries/| tries/' | cut -c1-330
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary | python tools/bench_brief.py headline
timeout 300 python tools/gpu_single_scene.py 2>&1 | tail -2
