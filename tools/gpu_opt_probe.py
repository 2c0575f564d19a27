"""development (round 5): A/B of solver options on the steps of one closed loop, teacher-forced -- the loop is driven by the reference options (the
round-4 iteration: bulk_rounds=0, cold), every option set then solves the SAME recorded step inputs: agreement of outputs / statuses / ladder counts
with the reference, iterations, device time per step.
usage: python tools/gpu_opt_probe.py WORKLOAD STEPS "name=v,name=v" ["name=v" ...]     (WORKLOAD: C4 C3 C5 C2b all3 bound2; option warm=1: dmpc_warm_start)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
which, steps = sys.argv[1], int(sys.argv[2])
optsets = [a for a in sys.argv[3:] if not a.startswith("--dump=")]
dump = [a[7:] for a in sys.argv[3:] if a.startswith("--dump=")]   # --dump=file.npz: the reference run's outputs (A/B of two builds: tools/with_lib.py + np.array_equal)
cfgname, N, S, variant = {"C4": ("C4", 10000, 1, "bound"), "C3": ("C3", 1000, 16, "softall"), "C5": ("C5", 200, 64, "repair"),
                          "C2b": ("C4", 100, 512, "bound"), "all3": ("C4", 100, 128, "all3"), "bound2": ("C4", 100, 128, "bound2"),
                          "cpp": ("C4", 100, 128, "cpp")}[which]
cfg = dict(wl.CONFIGS[cfgname]); kw = wl.solver_kwargs(cfg, N)
po, pf = wl.make_scenes(cfg, S, N, wl.SEED0 + 4)
def ctx(opts):
    d = mp.Dmpc(variant, **kw)
    warm = False
    for o in [x for x in opts.split(",") if x]:
        k, v = o.split("=")
        if k == "warm": warm = int(v) != 0
        else: d.debug_option(k, int(v))
    if warm: d.warm_start(True)
    d.profile(True)
    return d
ref = ctx("")
l, _, _ = ref.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
inputs, refs = [], []
for k in range(steps):
    inputs.append((l, xp, xv, xa))
    o = ref.step_batch(l, xp, xv, xa, pf); o["solve_ms"], o["scan_ms"], _ = ref.profile_read2()
    refs.append(o)
    ok = (o["status"] & 1) == 1
    l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
print(f"{which}: {S} x {N} agents of {variant}; reference = no options")
print("step  " + "  ".join(f"[{o}]" for o in ["reference"] + optsets))
tot = np.zeros(len(optsets) + 1)
ds = [ctx(o) for o in optsets]
for k in range(steps):
    c = refs[k]
    line = f"{k + 2:3d}  it {c['info'][..., 4].mean():5.1f}/{c['info'][..., 4].max():3d} {c['solve_ms']:.3f} ms"
    tot[0] += c["solve_ms"]
    for j, d in enumerate(ds):
        w = d.step_batch(*inputs[k], pf); sv, sc, _ = d.profile_read2()
        tot[j + 1] += sv
        ok = (c["status"] & 1) == 1
        dp = np.abs(c["p"] - w["p"])[ok].max() if ok.any() else 0.0
        same = np.array_equal(c["status"], w["status"]) and np.array_equal(c["info"][..., :4], w["info"][..., :4])
        line += f"  | it {w['info'][..., 4].mean():5.1f}/{w['info'][..., 4].max():3d} {sv:.3f} ms dp {dp:.1e} {'same' if same else 'DIFF'}"
        if not same:
            bad = np.argwhere((c["status"] != w["status"]) | (c["info"][..., :4] != w["info"][..., :4]).any(-1))[:3]
            for b in bad: line += f"\n        at {tuple(b)}: ref st {c['status'][tuple(b)]} info {c['info'][tuple(b)]} | st {w['status'][tuple(b)]} info {w['info'][tuple(b)]}"
    print(line)
if dump:
    np.savez(dump[0], **{f"{k}_{i}": refs[i][k] for i in range(steps) for k in ("p", "v", "a", "status", "info")})
print("solve ms, sum over the steps: " + "  ".join(f"{t:.3f}" for t in tot))
