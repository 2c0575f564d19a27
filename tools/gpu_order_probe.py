"""development aid: what would a better solve-launch order give on the hardware?  Forces launch orders through the
dmpc_debug_set_order hook: natural, random, by the TRUE iteration counts (upper bound of any predictor), by row count and
by the number of horizon steps with a row violated at the unconstrained minimiser."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import workload as wl
import bench
cfg = wl.CONFIGS["C2"]; N = 100; SB = 64; REP = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kw = wl.solver_kwargs(cfg, N)
d = mp.Dmpc("hard", **kw)
l, xp, xv, xa, pf, alive = bench.capture_state(d, cfg, SB, N, 12, wl.SEED0 + 2)
out = d.step_batch(l, xp, xv, xa, pf)
it = out["info"][..., 4].reshape(-1).astype(float); nr = out["info"][..., 1].reshape(-1).astype(float)
# feature: horizon steps with a row violated at the unconstrained minimiser
Lam, Av, A0, Dl = mp.model_matrices(kw["h"]); K = 15
q, s_ = kw["Q1"], kw["S1"]
Q = np.zeros((45, 45)); Q[42:, 42:] = q * np.eye(3)
H = 2 * (Lam.T @ Q @ Lam + s_ * Dl.T @ Dl + np.eye(45)); Hi = np.linalg.inv(H)
sv = np.zeros(SB * N); mmin = np.ones(SB * N)
for s in range(SB):
    for n in range(N):
        x0 = np.r_[xp[s, n], xv[s, n]]
        a01 = np.r_[xa[s, n], np.zeros(42)]
        f = -2 * ((np.tile(pf[s, n], K) - A0 @ x0) @ Q @ Lam + a01 @ (s_ * Dl))
        wu = (Lam @ (-Hi @ f)).reshape(K, 3)
        r = d.rows_one(l[s], n, xp[s, n], xv[s, n])
        if len(r["kc"]):
            kc = r["kc"] - 1
            viol = -(r["xi"] * wu[kc]).sum(1) - r["rhs"]
            sv[s * N + n] = len(set(kc[viol > 1e-10]))
            hw = 0.5 * kw["alim"] * ((kc + 1) * kw["h"]) ** 2
            rng_ = hw * np.abs(r["xi"]).sum(1)
            mmin[s * N + n] = ((r["rhs"] + rng_) / (2 * rng_)).min()
print("corr with iterations: rows %.2f  steps_viol %.2f  -m_min %.2f" % (np.corrcoef(nr, it)[0, 1], np.corrcoef(sv, it)[0, 1], np.corrcoef(-mmin, it)[0, 1]))
S = SB * REP
rep = lambda a: np.ascontiguousarray(np.concatenate([a] * REP, axis=0))
L_, XP, XV, XA, PF = rep(l), rep(xp), rep(xv), rep(xa), rep(pf)
T = S * N
tile = lambda v: np.concatenate([v] * REP)
Lb = d._L
Lb.dmpc_debug_set_order.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
Lb.dmpc_profile.argtypes = [C.c_void_p, C.c_int]
Lb.dmpc_profile_read2.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
rng = np.random.default_rng(1)
def run(name, order):
    if order is None: Lb.dmpc_debug_set_order(d._ctx, None, 0)
    else:
        o = np.ascontiguousarray(order, dtype=np.int32)
        Lb.dmpc_debug_set_order(d._ctx, o.ctypes.data_as(C.POINTER(C.c_int)), T)
    for _ in range(2): d.step_batch(L_, XP, XV, XA, PF)
    Lb.dmpc_profile(d._ctx, 1)
    for _ in range(6): res = d.step_batch(L_, XP, XV, XA, PF)
    a, b, n = C.c_double(), C.c_double(), C.c_int64()
    Lb.dmpc_profile_read2(d._ctx, C.byref(a), C.byref(b), C.byref(n))
    Lb.dmpc_profile(d._ctx, 0)
    print(f"{name:34s} solve {a.value*1e3:8.1f} us   scan(+order) {b.value*1e3:7.1f} us")
    return res
ref = run("natural (built-in policy)", None)
run("identity forced", np.arange(T))
run("random permutation", rng.permutation(T))
run("true iterations (ideal)", np.argsort(-tile(it), kind="stable"))
run("rows", np.argsort(-tile(nr), kind="stable"))
run("steps_viol", np.argsort(-tile(sv), kind="stable"))
run("steps_viol - 3 m_min", np.argsort(-(tile(sv) - 3 * tile(mmin)), kind="stable"))
k2 = tile(sv) * 4 + ((tile(mmin) < 0.1) * 3 + ((tile(mmin) >= 0.1) & (tile(mmin) < 0.25)) * 2 + ((tile(mmin) >= 0.25) & (tile(mmin) < 0.5)) * 1)
run("steps_viol*4 + tightness class", np.argsort(-k2, kind="stable"))
tc = (tile(mmin) < 0.1) * 3 + ((tile(mmin) >= 0.1) & (tile(mmin) < 0.25)) * 2 + ((tile(mmin) >= 0.25) & (tile(mmin) < 0.5)) * 1
svt, nrt = tile(sv), tile(nr)
for name, key in (("sv*4+tc, rows tie-break", (svt * 4 + tc) * 1000 + nrt), ("sv*8 + tc*2 + (rows>150)", svt * 8 + tc * 2 + (nrt > 150)),
                  ("sv*3 + tc*2", svt * 3 + tc * 2), ("sv*2 + tc", svt * 2 + tc), ("sv*4+tc + rows/40", svt * 4 + tc + nrt / 40.0),
                  ("sv + 4*(1-mmin)", svt + 4 * (1 - tile(mmin))), ("sv + 8*(1-mmin)", svt + 8 * (1 - tile(mmin))), ("sv^1.5*4+tc", svt ** 1.5 * 4 + tc)):
    run(name, np.argsort(-key, kind="stable"))
res = run("natural again", None)
assert np.array_equal(res["status"], ref["status"])
