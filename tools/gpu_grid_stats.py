"""development (round 4): what the cell-grid neighbour pass of a 10^4-agent scene has to look at -- a numpy model of grid_bin / grid_query on the
real predictions of MPC step 6: candidates per agent (cell ranges), survivors of the segment-box test, final list length (fp32 distance test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
cfg, N = dict(wl.CONFIGS["C4"]), 10000
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("bound", **kw)
po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 4)
l, _, _ = d.init_batch(po, pf)
xp, xv, xa = po.copy(), np.zeros_like(po), np.zeros_like(po)
for k in range(5):
    o = d.step_batch(l, xp, xv, xa, pf)
    ok = o["status"] == 1
    l = np.where(ok[..., None], o["p"], l); xp = np.where(ok[..., None], o["p"][..., :3], xp)
    xv = np.where(ok[..., None], o["v"][..., :3], xv); xa = np.where(ok[..., None], o["a"][..., :3], xa)
P = l[0].reshape(N, 15, 3)
c = cfg["c"]; R = 3 * cfg["rmin"] * 1.0001 + 1e-4; Rz = R * c
pmin, pmax = np.array(kw["pmin"]), np.array(kw["pmax"])
for cells in ((R, 1.5 * R, 1.5 * Rz), (R, R, Rz), (0.5 * R, R, Rz), (R, 2 * R, 2 * Rz)):
    n = np.clip(((pmax - pmin) / np.array(cells)).astype(int), 1, 32)
    inv = n / (pmax - pmin)
    tot_c = np.zeros(N); tot_runs = np.zeros(N); tot_rounds = np.zeros(N)
    for sg in range(3):
        seg = P[:, 5 * sg:5 * sg + 5]
        lo, hi = seg.min(1), seg.max(1)
        half = 0.5 * (hi - lo); ctr = 0.5 * (lo + hi)
        mh = half.max(0)
        cell = np.clip(np.floor((ctr - pmin) * inv).astype(int), 0, n - 1)
        H = np.zeros(tuple(n)); np.add.at(H, (cell[:, 0], cell[:, 1], cell[:, 2]), 1)
        reach = np.array([R, R, Rz]) + mh
        clo = np.clip(np.floor((lo - reach - pmin) * inv).astype(int), 0, n - 1)
        chi = np.clip(np.floor((hi + reach - pmin) * inv).astype(int), 0, n - 1)
        # prefix sums for box counts
        Cs = np.zeros(tuple(n + 1)); Cs[1:, 1:, 1:] = H.cumsum(0).cumsum(1).cumsum(2)
        def box(a, b):
            x0, y0, z0 = a.T; x1, y1, z1 = (b + 1).T
            return (Cs[x1, y1, z1] - Cs[x0, y1, z1] - Cs[x1, y0, z1] - Cs[x1, y1, z0] + Cs[x0, y0, z1] + Cs[x0, y1, z0] + Cs[x1, y0, z0] - Cs[x0, y0, z0])
        cand = box(clo, chi)
        runs = (chi[:, 1] - clo[:, 1] + 1) * (chi[:, 2] - clo[:, 2] + 1)
        tot_c += cand; tot_runs += runs; tot_rounds += runs + cand / 64.0    # (about one partial round per run on top of the full ones)
        if cells[1] == 1.5 * R: print(f"  segment {sg}: largest half extents {np.round(mh, 2)}, mean half {np.round(half.mean(0), 2)}")
    print(f"cells {np.round(cells, 2)} n {n}: candidates per agent mean {tot_c.mean():.0f} (max {tot_c.max():.0f}), x-runs {tot_runs.mean():.0f}, ~rounds {tot_rounds.mean():.0f}")
# survivors of the box test and final list length on a sample
rng = np.random.default_rng(0); sample = rng.choice(N, 200, replace=False)
e1 = np.array([1, 1, 1 / c])
surv, fin, chord, chord_box = [], [], [], []
Ps = P * e1   # (z scaled as the distance test scales it)
Rthr = np.sqrt((3 * cfg["rmin"]) ** 2 * 1.002)
def seg_chords(sg):
    seg = Ps[:, 5 * sg:5 * sg + 5]
    a0, a1 = seg[:, 0], seg[:, 4]
    t = np.arange(5) / 4.0
    lerp = a0[:, None] + t[None, :, None] * (a1 - a0)[:, None]
    dev = np.sqrt(((seg - lerp) ** 2).sum(-1)).max(1)
    return a0, a1, dev
CH = [seg_chords(sg) for sg in range(3)]
for i in sample:
    s_any = np.zeros(N, bool)
    for sg in range(3):
        seg = P[:, 5 * sg:5 * sg + 5]; lo, hi = seg.min(1), seg.max(1)
        inf = np.array([R, R, Rz])
        s_any |= ((lo <= hi[i] + inf) & (hi >= lo[i] - inf)).all(1)
    s_any[i] = False
    d2 = (((P - P[i]) * e1) ** 2).sum(-1)
    f = (d2 < (3 * cfg["rmin"]) ** 2 * 1.002).any(1); f[i] = False
    surv.append(s_any.sum()); fin.append(f.sum())
    # the time-synchronous chord test: per segment the closest approach of the two chords AT EQUAL TIME against R + both deviations
    c_any = np.zeros(N, bool)
    for sg in range(3):
        a0, a1, dev = CH[sg]
        r0 = a0 - a0[i]; dr = (a1 - a0) - (a1[i] - a0[i])
        den = (dr * dr).sum(1); tt = np.clip(-(r0 * dr).sum(1) / np.maximum(den, 1e-20), 0, 1)
        dmin = np.sqrt(((r0 + tt[:, None] * dr) ** 2).sum(1))
        c_any |= dmin <= Rthr + dev + dev[i] + 1e-4
    c_any[i] = False
    assert not (f & ~c_any).any(), "the chord test dropped a listed neighbour"
    chord.append(c_any.sum()); chord_box.append((c_any & s_any).sum())
print(f"time-synchronous chord test instead of the segment-box test: survivors per agent mean {np.mean(chord):.0f} max {np.max(chord)} (and inside the box test: {np.mean(chord_box):.0f}); deviation from the chord mean {np.mean([c[2].mean() for c in CH]):.3f} max {np.max([c[2].max() for c in CH]):.3f} m")
print(f"box-test survivors per agent mean {np.mean(surv):.0f} max {np.max(surv)}; final list mean {np.mean(fin):.1f} max {np.max(fin)}")
