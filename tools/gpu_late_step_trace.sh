REPO=$(pwd); OUT=$REPO/gpurun_out/kt_late; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -o a -- python $REPO/tools/gpu_bound_ab.py bound f64 70 > $OUT/a.log 2>&1
DMPC_DEBUG_OPTIONS=cull_min=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -o b -- python $REPO/tools/gpu_bound_ab.py bound f64 70 > $OUT/b.log 2>&1
cd $REPO
python3 - <<'PY'
import csv
for t in "ab":
    rows=list(csv.DictReader(open(f"gpurun_out/kt_late/{t}/{t}_kernel_trace.csv")))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    last=rows[-8:]
    t0=int(last[0]["Start_Timestamp"])
    print("==",t)
    for r in last:
        print(f'{(int(r["Start_Timestamp"])-t0)/1e3:8.1f} +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f} us  {r["Kernel_Name"][:70]}')
PY
