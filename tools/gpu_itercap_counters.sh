#!/bin/bash
# development (through gpurun): instruction counts of the headline solve launch under iteration caps (set-up + outputs; + crash batch; + n iterations at small q)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/itercap_counters; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; local opts=""; for o in "$@"; do opts="$opts --debug-option $o"; done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/$name" -o $name -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 9 --warmup 0 $opts > "$OUT/$name.log" 2>&1; }
run a iter_cap=0 crash_min=99
run b iter_cap=0
run c iter_cap=4 crash_min=99
run d iter_cap=8 crash_min=99
run e iter_cap=2000
cd "$REPO"; find "$OUT" -name "*.db" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "solve_persist_kernel<true, 56" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-2], {c: round(v / n[c] / 10000, 1) for c, v in acc.items()})
PY
