"""development (CPU): which launch-order key would have ordered the C4 solve queue best?  Input: gpurun_out/key_features.npz (tools/gpu_key_features.py).
List scheduling of the agents' work estimates on M wave slots in the order a candidate key gives; makespan against the perfect order."""
import sys, heapq
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/key_features.npz")["info"]   # [steps][N][8]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
def makespan(dur, order):
    h = [0.0] * M
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + dur[i])
    return max(h)
def feats(inf):
    w = inf[:, 5]
    key = w & 255; fin = (w >> 8) & 1
    qs = ((w >> 9) & 31) / 31.0; nsat = (w >> 14) & 63; nrv = (w >> 20) & 63; ls = (w >> 26) & 7
    return dict(key=key, fin=fin, cut=nsat.astype(float), smin=qs, ntight=nrv, ls=ls, nsat=nsat, nrv=nrv, rows=inf[:, 1], violk=inf[:, 0], cost=inf[:, 3] * 0.25 * 1.73, iters=inf[:, 4], tries=inf[:, 2], maxq=inf[:, 7])
S = [feats(d[k]) for k in range(d.shape[0])]
def evaluate(fn, label):
    ms = []
    for k in range(1, len(S)):
        f = S[k]; live = np.where(f["fin"] == 0)[0]
        kv = fn(f, S[k - 1])[live]
        order = live[np.argsort(-kv, kind="stable")]
        ms.append(makespan(f["cost"], order))
    print(f"{label:60s} mean makespan {np.mean(ms):7.1f} us   per step {np.round(ms).astype(int)}")
    return np.mean(ms)
evaluate(lambda f, p: f["cost"], "perfect (by the work estimate itself)")
evaluate(lambda f, p: np.random.default_rng(0).random(len(f["cost"])), "random")
evaluate(lambda f, p: f["key"].astype(float), "the key as built")
evaluate(lambda f, p: f["rows"].astype(float), "reference row count")
evaluate(lambda f, p: np.maximum(f["key"], p["cost"] / 4.0), "max(key, previous cost / 4 us)")
evaluate(lambda f, p: p["cost"], "previous cost")
evaluate(lambda f, p: -f["smin"], "smallest share")
evaluate(lambda f, p: f["nsat"].astype(float), "bounds violated at the unconstrained minimiser")
evaluate(lambda f, p: f["nsat"] + 2.0 * f["nrv"], "bounds + 2 x rows violated at the unconstrained minimiser")
evaluate(lambda f, p: f["nrv"].astype(float), "rows violated at the unconstrained minimiser")
evaluate(lambda f, p: f["ls"] * 100.0 + f["cut"], "ladder start, cut sum")
print("work per slot:", [round(S[k]["cost"].sum() / M) for k in range(1, len(S))], " longest:", [round(S[k]["cost"].max()) for k in range(1, len(S))])
# who are the heavy ones?
f = S[min(4, len(S) - 1)]; top = np.argsort(f["cost"])[-200:]
for name in ("rows", "smin", "nsat", "nrv", "ls", "tries", "maxq", "iters"):
    print(f"  heaviest 200 of one step: {name:7s} median {np.median(f[name][top]):.2f} (all agents {np.median(f[name]):.2f})  10th pct {np.percentile(f[name][top], 10):.2f}  90th {np.percentile(f[name][top], 90):.2f}")
# a linear search over simple combinations
best = None
rng = np.random.default_rng(1)
for trial in range(300):
    c = rng.random(6) * np.array([1.0, 60.0, 3.0, 8.0, 60.0, 0.5])
    fn = lambda f, p, c=c: c[0] * f["rows"] + c[1] * (1 - f["smin"]) + c[2] * f["cut"] + c[3] * f["ntight"] + c[4] * f["ls"] + c[5] * p["cost"]
    ms = []
    for k in range(1, len(S)):
        f = S[k]; live = np.where(f["fin"] == 0)[0]
        order = live[np.argsort(-fn(f, S[k - 1])[live], kind="stable")]
        ms.append(makespan(f["cost"], order))
    m = np.mean(ms)
    if best is None or m < best[0]: best = (m, c)
print("best linear combination of (rows, 1 - smin, nsat, nrv, ls, previous cost):", np.round(best[1], 2), "-> mean makespan", round(best[0], 1))
