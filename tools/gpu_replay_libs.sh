for n in base t3_1 t3_4 t3_6 t1_3; do
  for w in hard bound; do
    if [ $n = base ]; then r=$(python tools/replay_workload.py $w --steps 40 --warmup 5 2>/dev/null | tail -1); else r=$(python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_$n.so tools/replay_workload.py $w --steps 40 --warmup 5 2>/dev/null | tail -1); fi
    echo "$n $r" | cut -c1-110
  done
done
