# instructions per agent at two mean iteration counts (G=1: ~7.3, G=8 emulation: ~18) -> fixed part and per-iteration part
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for g in 0 8; do
  rm -rf /tmp/im$g
  EXTRA=""; [ $g -gt 0 ] && EXTRA="--emulate-gpus $g"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d /tmp/im$g -o im -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 $EXTRA > /tmp/im$g.log 2>&1
  tail -1 /tmp/im$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('G=$g', d['workload_stats']['mean_iters'], d['ms_per_step'])"
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/im$g/**/*counter_collection.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'solve_persist' in r['Kernel_Name']]
# per dispatch: group by Dispatch_Id
d=collections.defaultdict(dict)
for r in rows: d[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
last=list(d.values())[-3:]
for v in last: print({k:round(x/51200) for k,x in v.items()})
PY
done
