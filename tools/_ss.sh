python tools/gpu_single_scene.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/gpu_single_scene.py > /tmp/kt.log 2>&1
head -12 /tmp/kt/*kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/tmp/kt/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# a window of consecutive kernels in the middle of the bound transition
i0=len(rows)//3
prev=None
for r in rows[i0:i0+24]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'][:48].ljust(48), 'dur',(e-s)/1e3,'us gap',(s-prev)/1e3 if prev else 0)
    prev=e
PY
