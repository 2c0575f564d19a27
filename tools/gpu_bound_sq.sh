# development: instruction counters of the slack-variant persistent solve kernel in the bench's solveSoftDMPCbound replay
REPO=$(pwd); OUT=$REPO/gpurun_out/bound_sq; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT" -o sq -- python $REPO/bench.py --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/log.txt" 2>&1
cd $REPO
python3 - <<'PY'
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("gpurun_out/bound_sq/sq_counter_collection.csv")):
    k=r["Kernel_Name"]
    if "solve_persist_kernel<true, 48>" in k or "solve_persist_kernel<false, 48>" in k:
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k,v in acc.items():
    print(k[:60], "launches", n[k], {c: round(x/n[k]/51200,1) for c,x in v.items() if c!="SQ_WAVES"})
PY
