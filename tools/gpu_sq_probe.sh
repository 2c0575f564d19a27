#!/bin/bash
# development: VALU / scalar / LDS / branch instructions per agent of the C4 solve launch under launch-form options (rocprofv3 --pmc, one pass):
# the persistent split form (default), persistent unsplit, one agent per workgroup.   usage (through gpurun): bash tools/gpu_sq_probe.sh
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for opt in "" "no_split_t=1,force_persist=1" "no_split_t=1"; do
  rm -rf /tmp/sqp; DMPC_DEBUG_OPTIONS="$opt" timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d /tmp/sqp -o sq -- python $R/tools/replay_workload.py c4 --steps 8 --warmup 0 > /tmp/sqp.log 2>&1
  python3 - "$opt" <<'PY'
import csv,sys,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/sqp/**/sq_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'solve' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    n=len(d['SQ_INSTS_VALU'])
    if sum(d['SQ_INSTS_VALU'])/n < 1e6: continue
    print(f"[{sys.argv[1]}] {k}: per agent VALU {sum(d['SQ_INSTS_VALU'])/n/1e4:.0f} SALU {sum(d['SQ_INSTS_SALU'])/n/1e4:.0f} LDS {sum(d['SQ_INSTS_LDS'])/n/1e4:.0f} BR {sum(d['SQ_INSTS_BRANCH'])/n/1e4:.0f} waves {sum(d['SQ_WAVES'])/n:.0f}")
PY
done
