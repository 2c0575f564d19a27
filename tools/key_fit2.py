"""development (CPU): launch-order keys against MEASURED solve durations (gpurun_out/key_data_*.npz from tools/gpu_key_data.py): list scheduling on the launch's wave slots (KEY_SLOTS)
in the order a candidate key gives; makespan against the perfect order and the key as built."""
import sys, heapq, glob
import numpy as np
import os
M = int(os.environ.get('KEY_SLOTS', '2048'))   # wave slots of the solve launch (reduced solver: 256 CUs x 8; the general split-factor kernel: 1792)
files = sorted(glob.glob("gpurun_out/key_data_*.npz"))
def makespan(dur, order):
    h = [0.0] * M; heapq.heapify(h)
    for i in order: heapq.heappush(h, heapq.heappop(h) + dur[i])
    return max(h)
def feats(inf, pl):
    w = inf[:, 5]
    return dict(key=(w & 255).astype(float), fin=(w >> 8) & 1, smin=((w >> 9) & 31) / 31.0, nsat=((w >> 14) & 63).astype(float), nrv=((w >> 20) & 63).astype(float), ls=((w >> 26) & 7).astype(float),
                rows=pl[:, 1].astype(float), violk=pl[:, 0].astype(float), iters=pl[:, 4].astype(float), tries=pl[:, 2].astype(float), maxq=pl[:, 7].astype(float))
S = []
for f in files:
    d = np.load(f)
    for k in range(d["info"].shape[0]):
        ft = feats(d["info"][k], d["plain"][k]); ft["dur"] = d["dur"][k]; ft["file"] = f; ft["step"] = k + 2
        S.append(ft)
def evaluate(fn, label, sel=lambda s: s["step"] >= 3):
    ms = []
    for s in S:
        if not sel(s): continue
        live = np.where(s["dur"] > 0)[0]
        order = live[np.argsort(-fn(s)[live], kind="stable")]
        ms.append(makespan(s["dur"], order))
    print(f"{label:70s} mean makespan {np.mean(ms):7.1f} us  per step {np.round(ms[:8]).astype(int)}")
evaluate(lambda s: s["dur"], "perfect")
evaluate(lambda s: np.random.default_rng(0).random(len(s["dur"])), "random")
evaluate(lambda s: s["key"], "the key as built")
evaluate(lambda s: s["iters"], "iterations (hindsight)")
evaluate(lambda s: s["rows"], "rows")
evaluate(lambda s: s["nrv"], "rows violated at the unconstrained minimiser")
evaluate(lambda s: s["nsat"], "bounds violated at the unconstrained minimiser")
# least squares on the live agents of the training file(s), tested on all
names = ["rows", "smin", "nsat", "nrv", "ls", "violk"]
def design(s, extra=True):
    X = [np.ones_like(s["dur"])] + [s[n] for n in names]
    if extra:
        X += [s["rows"] * (1 - s["smin"]), s["nrv"] ** 2, s["rows"] ** 2, (1 - s["smin"]) ** 2, s["nsat"] * s["nrv"], np.minimum(s["rows"], 20), (s["violk"] <= 3).astype(float), s["nsat"] ** 2, s["rows"] * s["nrv"]]
    return np.stack(X, axis=1)
for extra in (False, True):
    tr = [s for s in S if s["file"] == files[0] and s["step"] >= 3]
    X = np.concatenate([design(s, extra)[s["dur"] > 0] for s in tr]); y = np.concatenate([s["dur"][s["dur"] > 0] for s in tr])
    coef, *_ = np.linalg.lstsq(X, y, rcond=None)
    pred = X @ coef
    print("extra terms" if extra else "linear", "R^2 on training", round(1 - ((y - pred) ** 2).sum() / ((y - y.mean()) ** 2).sum(), 3), "coef", np.round(coef, 2))
    evaluate(lambda s: design(s, extra) @ coef, f"least squares fit ({'with' if extra else 'no'} extra terms), test file", sel=lambda s: s["step"] >= 3 and s["file"] != files[0])
    evaluate(lambda s: s["key"], "the key as built, test file", sel=lambda s: s["step"] >= 3 and s["file"] != files[0])
k = S[2]; live = k["dur"] > 0
print("correlations with the duration (step 4):", {n: round(float(np.corrcoef(k[n][live], k["dur"][live])[0, 1]), 3) for n in names + ["key", "iters", "maxq", "tries"]})
print("work per slot", round(k["dur"].sum() / M), "longest", round(k["dur"].max()))
# the agent's own previous solve (closed loop): alone, and combined with the key as built
by = {(s["file"], s["step"]): s for s in S}
def prev_of(s, what):
    p = by.get((s["file"], s["step"] - 1))
    return p[what] if p is not None else np.zeros_like(s["dur"])
evaluate(lambda s: prev_of(s, "dur"), "previous MPC step's duration")
evaluate(lambda s: prev_of(s, "iters"), "previous MPC step's iterations / equality solves")
for w in (0.5, 1.0, 2.0):
    evaluate(lambda s: prev_of(s, "iters") * w + s["key"], f"{w} x previous iterations + key")
