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
import json, subprocess
summ = json.load(open(f"{src}/summary.json"))
try:   # the commit the passes were taken at (this script runs where the repository is; the GPU box has no .git)
    summ["git_commit"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=dst).stdout.strip()
    summ["git_dirty"] = bool(subprocess.run(["git", "status", "--porcelain", "--", "multiagent_planning_amd/csrc"], capture_output=True, text=True, cwd=os.path.dirname(dst)).stdout.strip())
except Exception:
    pass
json.dump(summ, open(f"{dst}/{prefix}_pmc_summary.json", "w"), indent=1)
fn = f"{src}/calib/calib_counter_collection.csv"
if os.path.exists(fn):
    rows = list(csv.DictReader(open(fn)))
    with open(f"{dst}/{prefix}_pmc_calib_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(r for r in rows if "read_probe" in r["Kernel_Name"])
print("collected", src, "->", prefix)
