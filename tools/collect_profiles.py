#!/usr/bin/env python3
"""Copies the passes of one tools/gpu_profile_round.sh run (gpurun_out/<label>/) into profiles/ under a round prefix:
   python tools/collect_profiles.py gpurun_out/<label> r03[_bound|_c4]
kernel stats as <prefix>_kernel_stats.csv (headline: <round>_bench_kernel_stats.csv), the counter CSVs as <prefix>_pmc_<pass>_counter_collection.csv
with the rows of this library's kernels only, the summary as <prefix>_pmc_summary.json."""
import csv, os, shutil, sys
src, prefix = sys.argv[1].rstrip("/"), sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
headline = prefix.count("_") == 0
keep = lambda name: any(t in name for t in ("dmpc", "pc::", "mg::"))
rows = list(csv.DictReader(open(f"{src}/kt/kt_kernel_stats.csv")))
out = f"{dst}/{prefix}_bench_kernel_stats.csv" if headline else f"{dst}/{prefix}_kernel_stats.csv"
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(r for r in rows if keep(r["Name"]))
for p in ("fetch", "write", "sq", "sq2", "sq3"):
    fn = f"{src}/{p}/{p}_counter_collection.csv"
    if not os.path.exists(fn): print("missing", fn); continue
    rows = list(csv.DictReader(open(fn)))
    with open(f"{dst}/{prefix}_pmc_{p}_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(r for r in rows if keep(r["Kernel_Name"]))
shutil.copy(f"{src}/summary.json", f"{dst}/{prefix}_pmc_summary.json")
print("collected", src, "->", prefix)
