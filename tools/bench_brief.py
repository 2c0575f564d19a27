#!/usr/bin/env python3
"""development: one line of the key numbers of a bench.py JSON line read from stdin (label = argv[1:])"""
import json, sys
txt = [l for l in sys.stdin.read().split("\n") if l.startswith("{")]
d = json.loads(txt[-1])
r, w = d["roofline"], d["workload_stats"]
print(" ".join(sys.argv[1:]), f"{d['value'] / 1e6:.2f} M/s  step {d['ms_per_step']:.3f} ms  solve {r['kernel_ms_avg']:.3f} ms  other {list(r['other_kernels_ms_avg'].values())[0]:.3f} ms"
      f"  iters {w['mean_iters']:.2f} solved {w['solved_frac']:.4f} infeas {w['infeasible_frac']:.4f} invalid {w['invalid']}")
