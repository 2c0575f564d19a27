#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/gpu_profile_round.sh into one JSON (stdout): per kernel and launch the counter
averages; for the solve kernel the HBM-side traffic per launch, the instruction mix per solve and the issue fractions the bench
line reports next to the HBM roofline.  usage: profile_summary.py gpurun_out/profile_round [solves_per_launch]"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 51200
CLOCK_HZ = 2.4e9            # MI355X peak engine clock (/opt/skills/guides/MI355X_MICROARCH.md); the counters are in cycles
N_SIMD = 256 * 4            # 256 CUs x 4 SIMDs


def short(n):
    return n.split("(")[0].replace("void ", "").split("<")[0].replace("dmpc::", "")


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: dict({c: sum(v) / len(v) for c, v in d.items()}, launches=max(len(v) for v in d.values())) for k, d in acc.items()}


passes = {}
for name in ("fetch", "write", "sq", "sq2"):
    p = os.path.join(root, name, f"{name}_counter_collection.csv")
    if os.path.exists(p):
        passes[name] = counters(p)
out = {"solves_per_launch": spl, "source": "tools/gpu_profile_round.sh (rocprofv3: --kernel-trace --stats; separate --pmc passes FETCH_SIZE, WRITE_SIZE, "
       "two SQ sets) on `python bench.py --no-cpu-baseline --no-secondary`"}
kt = os.path.join(root, "kt", "kt_kernel_stats.csv")
stats = {}
if os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
out["kernel_stats"] = {k: v for k, v in stats.items() if k.startswith("dmpc") or k in ("order_kernel", "bbox_kernel")}
KIB = 1024.0
traffic = {}
for k in set(passes.get("fetch", {})) | set(passes.get("write", {})):
    f = passes.get("fetch", {}).get(k, {}).get("FETCH_SIZE")
    w = passes.get("write", {}).get(k, {}).get("WRITE_SIZE")
    if f is not None or w is not None:
        traffic[k] = {"fetch_kib": f, "write_kib": w, "bytes": ((f or 0) + (w or 0)) * KIB}
out["traffic_per_launch"] = {k: v for k, v in traffic.items() if k.startswith("dmpc") or k == "order_kernel"}
sk = next((k for k in traffic if "solve_persist" in k), None)
if sk:
    step_kernels = [k for k in traffic if k.startswith("dmpc_s") or k == "order_kernel"]
    out["solve_kernel"] = sk
    out["hbm_bytes_per_launch"] = traffic[sk]["bytes"]
    out["whole_step_bytes_per_launch"] = sum(traffic[k]["bytes"] for k in step_kernels)
    out["algorithmic_bytes_per_launch"] = 1556.0 * spl
    out["traffic_calibration"] = ("WRITE_SIZE is exact in KiB on a known coalesced write (table_from_rows_kernel: 18000 KiB for 18,432,000 bytes); FETCH_SIZE is "
                                  "used raw (the guide's x2 correction is for 16-B/lane streams; these kernels read 8 B per lane)")
    sq, sq2 = passes.get("sq", {}).get(sk, {}), passes.get("sq2", {}).get(sk, {})
    dur_s = stats.get(sk, {}).get("avg_us", 0) * 1e-6
    if sq and dur_s:
        mix = {c.replace("SQ_INSTS_", "").lower(): sq[c] / spl for c in sq if c.startswith("SQ_INSTS")}
        total = sum(sq[c] for c in sq if c.startswith("SQ_INSTS"))
        out["instructions_per_solve"] = dict(mix, total=total / spl)
        slots = N_SIMD * dur_s * CLOCK_HZ / 4.0                       # one wave instruction per SIMD and quad-cycle
        out["issue"] = {"issue_slot_frac": total / slots,
                        "fp64_valu_frac": sq.get("SQ_INSTS_VALU", 0) / slots,   # VALU wave-instructions against one fp64 FMA per lane, SIMD and cycle
                        "lds_frac": sq.get("SQ_INSTS_LDS", 0) / slots,
                        "note": f"slots = {N_SIMD} SIMDs x kernel time x {CLOCK_HZ / 1e9:.1f} GHz / 4 (a wave64 instruction occupies a SIMD for 4 cycles); "
                                "the clock under load is lower than the peak used here, so the fractions are lower bounds"}
    if sq2:
        wc = sq2.get("SQ_WAVE_CYCLES", 0) or 1.0
        out["wave_time"] = {"active_any": sq2.get("SQ_ACTIVE_INST_ANY", 0) / wc, "active_valu": sq2.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                            "active_scalar": sq2.get("SQ_ACTIVE_INST_SCA", 0) / wc, "active_lds": sq2.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                            "waiting_on_counters": sq2.get("SQ_WAIT_ANY", 0) / wc, "waves": passes.get("sq", {}).get(sk, {}).get("SQ_WAVES")}
out["counters"] = {p: {k: v for k, v in d.items() if k.startswith("dmpc") or k == "order_kernel"} for p, d in passes.items()}
print(json.dumps(out, indent=1))
