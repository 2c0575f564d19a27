"""ctypes binding of the CPU oracle (oracle/dmpc_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under multiagent_planning_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdmpc_oracle.so")

VARIANTS = dict(bound=0, bound2=1, all3=2, hard=3, ondemand=4, ellip=5, softall=6, repair=7, cpp=8, cpp2=9, cpp1=10, softall_c=11, scp=12)
ST_SOLVED, ST_OUTBOUND, ST_COLL, ST_INFEAS = 1, 2, 4, 8
INFO_LEN = 8
I_VIOLK, I_NV, I_TRIES, I_CASE, I_ITERS, I_NSLACK, I_NACTIVE, I_NROWS = range(8)


class Params(C.Structure):
    _fields_ = [
        ("K", C.c_int), ("variant", C.c_int), ("order", C.c_int), ("max_tries", C.c_int),
        ("h", C.c_double), ("rmin", C.c_double), ("c", C.c_double), ("alim", C.c_double),
        ("Q1", C.c_double), ("S1", C.c_double), ("term", C.c_double),
        ("pmin", C.c_double * 3), ("pmax", C.c_double * 3),
        ("Qfar", C.c_double), ("Qnear", C.c_double), ("Sfree", C.c_double),
        ("tol", C.c_double),
    ]


def build(force=False, cflags=None, out=None):
    """Compile the oracle (gcc). Returns the path of the shared library."""
    out = out or _SO
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(
            os.path.join(_HERE, "dmpc_oracle.c")):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = ["gcc"] + (cflags or ["-O3", "-march=x86-64-v3"]) + [
            "-fPIC", "-std=c11", "-shared", "-o", out, os.path.join(_HERE, "dmpc_oracle.c"), "-lm", "-lpthread"]
        subprocess.check_call(cmd)
    return out


_lib = None


def lib(path=None):
    global _lib
    if _lib is None or path is not None:
        so = path or build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.orc_model_matrices.argtypes = [C.c_double, C.c_int, dp, dp, dp, dp]
        L.orc_init_one.argtypes = [dp, dp, C.c_double, C.c_int, dp, dp, dp]
        L.orc_solve_one.argtypes = [C.POINTER(Params), C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, dp]
        L.orc_step.argtypes = [C.POINTER(Params), C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, dp, C.c_int]
        L.orc_eval_one.argtypes = [C.POINTER(Params), C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_rows_one.argtypes = [C.POINTER(Params), C.c_int, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, ip, ip, ip]
        L.orc_qp_dense.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, ip]
        L.orc_assemble_one.argtypes = [C.POINTER(Params), C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_int, ip, ip, ip, dp, dp, dp, dp]
        L.orc_step_scenes.argtypes = [C.POINTER(Params), C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, C.c_int]
        if path is not None:
            return L
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_params(variant, K=15, h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4,
                pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2), max_tries=0, Qfar=0.0, Qnear=0.0, Sfree=0.0, order=2, tol=2.0):
    p = Params()
    p.K, p.variant, p.order, p.max_tries = K, VARIANTS[variant] if isinstance(variant, str) else variant, int(order), max_tries
    p.h, p.rmin, p.c, p.alim, p.Q1, p.S1, p.term = h, rmin, c, alim, Q1, S1, term
    p.Qfar, p.Qnear, p.Sfree = Qfar, Qnear, Sfree
    p.tol = float(tol)
    for i in range(3):
        p.pmin[i] = float(pmin[i])
        p.pmax[i] = float(pmax[i])
    return p


def model_matrices(h, K):
    n = 3 * K
    Lam, Av, A0, Dl = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, 6)), np.zeros((n, n))
    rc = lib().orc_model_matrices(h, K, _dp(Lam), _dp(Av), _dp(A0), _dp(Dl))
    assert rc == 0
    return Lam, Av, A0, Dl


def init_one(po, pf, h, K):
    po, pf = _f(po), _f(pf)
    p, v, a = np.zeros(3 * K), np.zeros(3 * K), np.zeros(3 * K)
    lib().orc_init_one(_dp(po), _dp(pf), h, K, _dp(p), _dp(v), _dp(a))
    return p, v, a


def solve_one(prm, l, n, po, vo, ao, pf):
    l = _f(l)
    N = l.shape[0]
    n3 = 3 * prm.K
    po, vo, ao, pf = _f(po), _f(vo), _f(ao), _f(pf)
    p, v, a = np.zeros(n3), np.zeros(n3), np.zeros(n3)
    info = np.zeros(INFO_LEN, dtype=np.int32)
    obj = C.c_double(0.0)
    st = lib().orc_solve_one(C.byref(prm), N, n, _dp(l), _dp(po), _dp(vo), _dp(ao), _dp(pf), _dp(p), _dp(v), _dp(a),
                             _ip(info), C.byref(obj))
    return dict(status=st, p=p, v=v, a=a, info=info, obj=obj.value)


def step(prm, l, x_p, x_v, x_a, pf, nthreads=1, library=None):
    l, x_p, x_v, x_a, pf = _f(l), _f(x_p), _f(x_v), _f(x_a), _f(pf)
    N = l.shape[0]
    n3 = 3 * prm.K
    p, v, a = np.zeros((N, n3)), np.zeros((N, n3)), np.zeros((N, n3))
    status = np.zeros(N, dtype=np.int32)
    info = np.zeros((N, INFO_LEN), dtype=np.int32)
    obj = np.zeros(N)
    L = library or lib()
    rc = L.orc_step(C.byref(prm), N, _dp(l), _dp(x_p), _dp(x_v), _dp(x_a), _dp(pf), _dp(p), _dp(v), _dp(a),
                    _ip(status), _ip(info), _dp(obj), nthreads)
    assert rc == 0
    return dict(status=status, p=p, v=v, a=a, info=info, obj=obj)


def eval_one(prm, l, n, po, vo, ao, pf, acc):
    l, po, vo, ao, pf, acc = _f(l), _f(po), _f(vo), _f(ao), _f(pf), _f(acc)
    obj, mv = C.c_double(0.0), C.c_double(0.0)
    rc = lib().orc_eval_one(C.byref(prm), l.shape[0], n, _dp(l), _dp(po), _dp(vo), _dp(ao), _dp(pf), _dp(acc),
                            C.byref(obj), C.byref(mv))
    return rc, obj.value, mv.value


def qp_dense(G, g, Cm, d):
    G, g, Cm, d = _f(G), _f(g), _f(Cm), _f(d)
    n, m = G.shape[0], Cm.shape[0]
    x, lam = np.zeros(n), np.zeros(max(m, 1))
    it = C.c_int(0)
    rc = lib().orc_qp_dense(n, m, _dp(G), _dp(g), _dp(Cm), _dp(d), _dp(x), _dp(lam), C.byref(it))
    return rc, x, lam[:m], it.value


def rows_one(prm, l, n, po, vo, max_rows=4096):
    """collision rows of agent n (dense): dict(G [nr,3K], b, dist, nrows, viol_k, status)."""
    l, po, vo = _f(l), _f(po), _f(vo)
    n3 = 3 * prm.K
    G, b, dist = np.zeros((max_rows, n3)), np.zeros(max_rows), np.zeros(max_rows)
    nr, vk, st = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().orc_rows_one(C.byref(prm), l.shape[0], n, _dp(l), _dp(po), _dp(vo), max_rows, _dp(G), _dp(b), _dp(dist),
                            C.byref(nr), C.byref(vk), C.byref(st))
    assert rc == 0
    k = min(nr.value, max_rows)
    return dict(G=G[:k], b=b[:k], dist=dist[:k], nrows=nr.value, viol_k=vk.value, status=st.value)


def assemble_one(prm, l, n, po, vo, ao, pf, level=0):
    """The literal dense QP of agent n at retry-ladder level `level`: dict(H, f, C, d, ncoll) with
    min 1/2 x'Hx + f'x s.t. Cx <= d, x = [a; eps]; None when the variant returns `coll` before building a QP."""
    l, po, vo, ao, pf = _f(l), _f(po), _f(vo), _f(ao), _f(pf)
    nn, mm, nc = C.c_int(0), C.c_int(0), C.c_int(0)
    null = C.POINTER(C.c_double)()
    rc = lib().orc_assemble_one(C.byref(prm), l.shape[0], n, _dp(l), _dp(po), _dp(vo), _dp(ao), _dp(pf), level,
                                C.byref(nn), C.byref(mm), C.byref(nc), null, null, null, null)
    if rc != 0:
        return None
    H, f, Cm, d = np.zeros((nn.value, nn.value)), np.zeros(nn.value), np.zeros((mm.value, nn.value)), np.zeros(mm.value)
    lib().orc_assemble_one(C.byref(prm), l.shape[0], n, _dp(l), _dp(po), _dp(vo), _dp(ao), _dp(pf), level,
                           C.byref(nn), C.byref(mm), C.byref(nc), _dp(H), _dp(f), _dp(Cm), _dp(d))
    return dict(H=H, f=f, C=Cm, d=d, ncoll=nc.value)


def step_scenes(prm, l, x_p, x_v, x_a, pf, nthreads=1):
    """S independent scenes ([S,N,...] arrays), scene-parallel over `nthreads` host threads (bench.py's CPU baseline)."""
    l, x_p, x_v, x_a, pf = _f(l), _f(x_p), _f(x_v), _f(x_a), _f(pf)
    S, N = l.shape[0], l.shape[1]
    n3 = 3 * prm.K
    p, v, a = np.zeros((S, N, n3)), np.zeros((S, N, n3)), np.zeros((S, N, n3))
    status = np.zeros((S, N), dtype=np.int32)
    info = np.zeros((S, N, INFO_LEN), dtype=np.int32)
    rc = lib().orc_step_scenes(C.byref(prm), S, N, _dp(l), _dp(x_p), _dp(x_v), _dp(x_a), _dp(pf), _dp(p), _dp(v), _dp(a),
                               _ip(status), _ip(info), nthreads)
    assert rc == 0
    return dict(status=status, p=p, v=v, a=a, info=info)
