#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/gpu_profile_round.sh: per kernel and launch the counter averages, and the
derived per-solve instruction mix of the solve kernel.  usage: profile_summary.py gpurun_out/profile_round [solves_per_launch]"""
import csv, sys, collections, json, os
root = sys.argv[1]
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 51200
def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("dmpc::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in acc.items()}
out = {}
for name in ("fetch", "write", "sq"):
    p = os.path.join(root, name, f"{name}_counter_collection.csv")
    if os.path.exists(p):
        out[name] = counters(p)
kt = os.path.join(root, "kt", "kt_kernel_stats.csv")
if os.path.exists(kt):
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in csv.DictReader(open(kt))][:6]
sq = out.get("sq", {})
for k, d in sq.items():
    if "solve" in k and "SQ_INSTS_VALU" in d:
        d["per_solve"] = {c: d[c] / spl for c in d if c.startswith("SQ_INSTS")}
        d["valu_active_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1)
print(json.dumps(out, indent=1))
