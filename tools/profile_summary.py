#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/gpu_profile_round.sh into one JSON (stdout): per kernel and launch the counter
averages; for the solve kernel the HBM-side traffic per launch, the instruction mix per solve and the issue fractions the bench
line reports next to the HBM roofline.  usage: profile_summary.py gpurun_out/profile_round [solves_per_launch] [kernel name substring]"""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 51200
kern_sub = sys.argv[3] if len(sys.argv) > 3 else "solve_persist"
waves_per_simd = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0    # resident waves per SIMD of the solve kernel (12 persistent waves per CU = 3)


def source_hash():
    """sha256 over the kernel sources the numbers belong to (bench.py refuses a summary whose hash is not the one of the sources it runs)"""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multiagent_planning_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]
CLOCK_HZ = 2.4e9            # MI355X peak engine clock (/opt/skills/guides/MI355X_MICROARCH.md); the counters are in cycles
N_SIMD = 256 * 4            # 256 CUs x 4 SIMDs


def short(n):   # template arguments kept: the slack-free and the slack-carrying solve kernels are different kernels
    return n.split("(")[0].replace("void ", "").replace("dmpc::", "").strip()


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: dict({c: sum(v) / len(v) for c, v in d.items()}, launches=max(len(v) for v in d.values())) for k, d in acc.items()}


passes = {}
for name in ("fetch", "write", "sq", "sq2", "sq3"):   # (+ calib: FETCH_SIZE of the read probe, handled below)
    p = os.path.join(root, name, f"{name}_counter_collection.csv")
    if os.path.exists(p):
        passes[name] = counters(p)
out = {"solves_per_launch": spl, "source_hash": source_hash(), "source": "tools/gpu_profile_round.sh (rocprofv3: --kernel-trace --stats; separate --pmc passes FETCH_SIZE, WRITE_SIZE, "
       "two SQ sets) on `python bench.py --no-cpu-baseline --no-secondary`"}
kt = os.path.join(root, "kt", "kt_kernel_stats.csv")
stats = {}
if os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
STEP_KERNELS = ("order_kernel", "bbox_kernel", "nbr_kernel", "table_nbrmajor_kernel", "grid_bin_kernel", "grid_scan_kernel", "grid_fill_kernel", "grid_query_kernel")
out["kernel_stats"] = {k: v for k, v in stats.items() if k.startswith("dmpc") or k.split("<")[0] in STEP_KERNELS}
KIB = 1024.0
# FETCH_SIZE calibration on this library's access pattern (tools/gpu_fetch_calib.py: 3 launches of a 512 MiB coalesced streaming read at 8 and
# at 16 bytes per lane): factor = known bytes / (counter x 1 KiB); applied to every FETCH_SIZE below (the kernels read 8 bytes per lane)
fetch_factor, calib = 1.0, None
cp = os.path.join(root, "calib", "calib_counter_collection.csv")
if os.path.exists(cp):
    c8, c16 = [], []
    for r in csv.DictReader(open(cp)):
        if "read_probe_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            (c16 if "double2" in r["Kernel_Name"] or "HIP_vector" in r["Kernel_Name"] else c8).append(float(r["Counter_Value"]))
    known = 512.0 * 1024 * 1024
    if c8 and c16:
        f8, f16 = known / (sum(c8) / len(c8) * KIB), known / (sum(c16) / len(c16) * KIB)
        fetch_factor = f8
        calib = {"bytes_per_launch": known, "fetch_size_kib_8B_per_lane": sum(c8) / len(c8), "fetch_size_kib_16B_per_lane": sum(c16) / len(c16),
                 "factor_8B_per_lane": f8, "factor_16B_per_lane": f16, "applied": "factor_8B_per_lane (tables, rows and states are read 8 bytes per lane)"}
traffic = {}
for k in set(passes.get("fetch", {})) | set(passes.get("write", {})):
    f = passes.get("fetch", {}).get(k, {}).get("FETCH_SIZE")
    w = passes.get("write", {}).get(k, {}).get("WRITE_SIZE")
    if f is not None or w is not None:
        traffic[k] = {"fetch_kib": f, "write_kib": w, "bytes": ((f or 0) * fetch_factor + (w or 0)) * KIB}
out["traffic_per_launch"] = {k: v for k, v in traffic.items() if k.startswith("dmpc") or k.split("<")[0] in STEP_KERNELS}
sk = max((k for k in traffic if kern_sub in k), key=lambda k: stats.get(k, {}).get("pct", 0), default=None)
if sk:
    step_kernels = [k for k in traffic if k.startswith("dmpc_s") or k.split("<")[0] in STEP_KERNELS]
    out["solve_kernel"] = sk
    out["hbm_bytes_per_launch"] = traffic[sk]["bytes"]
    out["whole_step_bytes_per_launch"] = sum(traffic[k]["bytes"] for k in step_kernels)
    out["algorithmic_bytes_per_launch"] = 1556.0 * spl
    out["traffic_calibration"] = calib if calib else ("WRITE_SIZE is exact in KiB on a known coalesced write (table_from_rows_kernel: 18000 KiB for 18,432,000 bytes); "
                                                      "FETCH_SIZE uncalibrated in this run (no calib pass): used raw")
    sq, sq2 = passes.get("sq", {}).get(sk, {}), passes.get("sq2", {}).get(sk, {})
    dur_s = stats.get(sk, {}).get("avg_us", 0) * 1e-6
    if sq and dur_s:
        mix = {c.replace("SQ_INSTS_", "").lower(): sq[c] / spl for c in sq if c.startswith("SQ_INSTS")}
        total = sum(sq[c] for c in sq if c.startswith("SQ_INSTS"))
        out["instructions_per_solve"] = dict(mix, total=total / spl)
        out["instruction_rate"] = {"wave_instructions_per_simd_and_cycle": total / (N_SIMD * dur_s * CLOCK_HZ),
                                   "note": f"all classes together, {N_SIMD} SIMDs, kernel time x {CLOCK_HZ / 1e9:.1f} GHz peak clock; NOT an issue-slot fraction: VALU, scalar, LDS and branch "
                                           "instructions of DIFFERENT waves issue side by side -- the pipe-level picture is `wave_time`"}
    sq3 = passes.get("sq3", {}).get(sk, {})
    if sq3 and sq and dur_s:
        f64 = {c: sq3.get("SQ_INSTS_VALU_" + c, 0.0) for c in ("FMA_F64", "ADD_F64", "MUL_F64", "TRANS_F64")}
        arith = sum(f64.values())
        valu = sq.get("SQ_INSTS_VALU", 0) or 1.0
        # lanes: SQ_THREAD_CYCLES_VALU accumulates the active lanes of every VALU instruction (64 for a full wave)
        lane_util = sq3.get("SQ_THREAD_CYCLES_VALU", 0.0) / (valu * 64.0) if sq3.get("SQ_THREAD_CYCLES_VALU") else None
        flops = (2.0 * f64["FMA_F64"] + f64["ADD_F64"] + f64["MUL_F64"]) * 64.0 * (lane_util or 1.0)
        out["fp64"] = {"valu_insts_per_solve": {k.lower(): v / spl for k, v in f64.items()}, "int32_per_solve": sq3.get("SQ_INSTS_VALU_INT32", 0.0) / spl,
                       "fp64_share_of_valu": arith / valu, "lane_utilisation": lane_util,
                       "achieved_tflops": flops / dur_s / 1e12, "peak_tflops_vector_fp64": 78.6,
                       "salu_cycles_per_solve": sq3.get("SQ_INST_CYCLES_SALU", 0.0) / spl, "lds_bank_conflict_cycles_per_solve": sq3.get("SQ_LDS_BANK_CONFLICT", 0.0) / spl,
                       "note": "fp64 arithmetic = FMA + ADD + MUL + TRANS wave-instructions; flops = (2 FMA + ADD + MUL) x 64 lanes x lane utilisation"}
    if sq2:
        wc = sq2.get("SQ_WAVE_CYCLES", 0) or 1.0
        out["wave_time"] = {"active_any": sq2.get("SQ_ACTIVE_INST_ANY", 0) / wc, "active_valu": sq2.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                            "active_scalar": sq2.get("SQ_ACTIVE_INST_SCA", 0) / wc, "active_lds": sq2.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                            "waiting_on_counters": sq2.get("SQ_WAIT_ANY", 0) / wc, "waiting_to_issue": sq2.get("SQ_WAIT_INST_ANY", 0) / wc,
                            "waves": passes.get("sq", {}).get(sk, {}).get("SQ_WAVES"), "resident_waves_per_simd": waves_per_simd,
                            "valu_pipe_busy": sq2.get("SQ_ACTIVE_INST_VALU", 0) / wc * waves_per_simd,
                            "note": "fractions of a wave's resident cycles (SQ_ACTIVE_INST_VALU / _SCA / _LDS, SQ_WAIT_ANY = parked on s_waitcnt, SQ_WAIT_INST_ANY = ready but "
                                    "not issued) ; valu_pipe_busy = active_valu x resident waves per SIMD"}
out["counters"] = {p: {k: v for k, v in d.items() if k.startswith("dmpc") or k.split("<")[0] in STEP_KERNELS} for p, d in passes.items()}
print(json.dumps(out, indent=1))
