"""ctypes binding of libdmpc_hip.so (the C ABI in include/dmpc_hip.h).

The shared library is built in-tree (multiagent_planning_amd/libdmpc_hip.so) by
`__graft_entry__.build()` / `make -C multiagent_planning_amd/csrc`.  There is no fallback of any
kind: if the library is missing, or no HIP device is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdmpc_hip.so")

VARIANTS = dict(bound=0, bound2=1, all3=2, hard=3, ondemand=4, ellip=5, softall=6, repair=7, cpp=8, cpp2=9, cpp1=10, softall_c=11, scp=12)
ST_SOLVED, ST_OUTBOUND, ST_COLL, ST_INFEAS, ST_CAPACITY, ST_ITERCAP = 1, 2, 4, 8, 16, 32
PRECISIONS = dict(f64=0, mixed=1, f32factor=2, low=3)
ST_REACHED = 256   # scene_status of transition(): every agent reached its goal
INFO_LEN = 8
I_VIOLK, I_NROWS, I_TRIES, I_CASE, I_ITERS, I_NSLACK, I_NACTIVE, I_MAXQ = range(8)
K_HOR = 15
ABI_VERSION = 7    # DMPC_ABI_VERSION of include/dmpc_hip.h

# every symbol include/dmpc_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "dmpc_create", "dmpc_destroy", "dmpc_last_error", "dmpc_set_params", "dmpc_model_matrices",
    "dmpc_posvel_matrix", "dmpc_init_batch", "dmpc_step_batch", "dmpc_solve_one", "dmpc_step_device",
    "dmpc_table_from_rows_device", "dmpc_advance_device", "dmpc_transition", "dmpc_solve_count",
    "dmpc_profile", "dmpc_profile_read", "dmpc_profile_read2", "dmpc_rows_one", "dmpc_postcheck",
    "dmpc_coll_rows", "dmpc_coll_rows_device", "dmpc_add_coll_constr", "dmpc_add_coll_constr_device",
    "dmpc_trajectories2file", "dmpc_test2file", "dmpc_random_test", "dmpc_random_exchange", "dmpc_random_sets_device",
    "dmpc_prop_state", "dmpc_is_inbounds", "dmpc_reached_goal", "dmpc_rows_dense",
    "dmpc_partition", "dmpc_comm_unique_id", "dmpc_comm_init", "dmpc_comm_destroy", "dmpc_step_sharded_device",
    "dmpc_transition_sharded", "dmpc_transition_sharded_gather", "dmpc_group_size", "dmpc_comm_size", "dmpc_abi_version", "dmpc_last_solve_kernel",
    "dmpc_max_deviation",
]


class DmpcParams(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("variant", C.c_int32), ("order", C.c_int32), ("max_tries", C.c_int32),
        ("h", C.c_double), ("rmin", C.c_double), ("c", C.c_double), ("alim", C.c_double),
        ("Q1", C.c_double), ("S1", C.c_double), ("term", C.c_double),
        ("pmin", C.c_double * 3), ("pmax", C.c_double * 3),
        ("Qfar", C.c_double), ("Qnear", C.c_double), ("Sfree", C.c_double),
        ("tol", C.c_double),
    ]


class DmpcError(RuntimeError):
    pass


_lib = None


def load():
    """Load libdmpc_hip.so; raises DmpcError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process only work if torch's is
    # loaded first (then this library binds to the same runtime by soname).  Import torch, when present,
    # before loading -- it is the device-memory / stream / RCCL plumbing of the callers anyway.
    try:
        import torch  # noqa: F401
    except Exception:   # torch is optional for the pure C-ABI use
        pass
    if not os.path.exists(LIB_PATH):
        raise DmpcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C multiagent_planning_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    # the header revision these bindings were written against (DMPC_ABI_VERSION: the special device values DEVICE_ALL / DEVICE_CURRENT below
    # changed once between revisions): refuse another library instead of passing it values that mean something else there
    # (DMPC_SKIP_ABI_CHECK=1: development A/B runs against an older build of the library, tools/with_lib.py)
    if not os.environ.get("DMPC_SKIP_ABI_CHECK") and (not hasattr(L, "dmpc_abi_version") or L.dmpc_abi_version() != ABI_VERSION):
        raise DmpcError(f"{LIB_PATH}: ABI revision {L.dmpc_abi_version() if hasattr(L, 'dmpc_abi_version') else '< 5'}, these bindings need {ABI_VERSION}: rebuild the library")
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    pp = C.POINTER(DmpcParams)
    L.dmpc_create.restype = vp
    L.dmpc_create.argtypes = [pp, C.c_int, C.c_int]
    L.dmpc_destroy.restype = None
    L.dmpc_destroy.argtypes = [vp]
    L.dmpc_last_error.restype = C.c_char_p
    L.dmpc_last_error.argtypes = [vp]
    L.dmpc_set_params.argtypes = [vp, pp]
    L.dmpc_model_matrices.argtypes = [pp, dp, dp, dp, dp]
    L.dmpc_posvel_matrix.argtypes = [C.c_double, C.c_int, dp]
    L.dmpc_init_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
    L.dmpc_step_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip]
    L.dmpc_solve_one.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip]
    L.dmpc_rows_one.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, ip, ip, ip, ip]
    L.dmpc_step_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 12
    L.dmpc_table_from_rows_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.dmpc_advance_device.argtypes = [vp, C.c_int] + [vp] * 8
    L.dmpc_transition.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_double, dp, dp, dp, ip, ip]
    i64 = C.c_int64
    L.dmpc_coll_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, dp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp,
                                 C.c_int, C.c_int, i64, i64, dp, i64, i64, dp, dp]
    L.dmpc_coll_rows_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, vp, i64, i64,
                                        C.c_int, vp, i64, i64, vp, vp, vp]
    L.dmpc_add_coll_constr.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp, C.c_int, i64, i64, dp, i64, i64, dp]
    L.dmpc_add_coll_constr_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_double, C.c_double, vp, i64, i64, C.c_int, vp, i64,
                                              i64, vp, vp]
    L.dmpc_trajectories2file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, dp, dp, dp, dp, dp]
    L.dmpc_test2file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, dp, dp, dp]
    L.dmpc_random_test.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, C.c_uint64, dp, dp]
    L.dmpc_random_exchange.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_uint64, dp, dp]
    L.dmpc_random_sets_device.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, C.c_uint64, C.c_int, vp, vp]
    L.dmpc_rows_dense.argtypes = [vp, C.c_int, dp, ip, dp, C.c_int, C.c_int, i64, i64, dp, i64, i64]
    L.dmpc_prop_state.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    L.dmpc_is_inbounds.argtypes = [vp, C.c_int, dp, dp, dp, ip]
    L.dmpc_reached_goal.argtypes = [vp, C.c_int, dp, dp, C.c_double, ip]
    L.dmpc_max_deviation.argtypes = [vp, C.c_int, dp, dp, dp]
    L.dmpc_postcheck.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, dp, dp, C.c_double, C.c_double, C.c_double,
                                 dp, dp, ip, dp, ip, dp, dp, dp, C.c_int]
    L.dmpc_partition.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, ip]
    L.dmpc_comm_unique_id.argtypes = [C.c_char_p]
    L.dmpc_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.dmpc_comm_destroy.argtypes = [vp]
    L.dmpc_comm_size.argtypes = [vp]
    L.dmpc_step_sharded_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 12
    L.dmpc_transition_sharded.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_double, dp, dp, dp, ip, ip]
    L.dmpc_transition_sharded_gather.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_double, dp, dp, dp, ip, ip]
    L.dmpc_group_size.argtypes = [vp]
    L.dmpc_debug_emulate_devices.argtypes = [C.c_int]
    L.dmpc_debug_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.dmpc_solve_count.restype = C.c_int64
    L.dmpc_solve_count.argtypes = [vp]
    L.dmpc_profile.argtypes = [vp, C.c_int]
    L.dmpc_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.dmpc_profile_read2.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.dmpc_last_solve_kernel.argtypes = [vp]
    L.dmpc_last_solve_kernel.restype = C.c_char_p
    _lib = L
    return L


def make_params(variant="bound", K=K_HOR, h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4,
                pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2), order=2, max_tries=0, Qfar=0.0, Qnear=0.0, Sfree=0.0, tol=2.0):
    p = DmpcParams()
    p.K, p.order, p.max_tries = int(K), int(order), int(max_tries)
    p.variant = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    p.h, p.rmin, p.c, p.alim, p.Q1, p.S1, p.term = float(h), float(rmin), float(c), float(alim), float(Q1), float(S1), float(term)
    for i in range(3):
        p.pmin[i] = float(pmin[i])
        p.pmax[i] = float(pmax[i])
    p.Qfar, p.Qnear, p.Sfree = float(Qfar), float(Qnear), float(Sfree)
    p.tol = float(tol)
    return p


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _strided_base(A):
    """(pointer, row stride, column stride) in elements of a 2-D float64 array with positive strides (C or Fortran order,
    or a view): the ABI's A[i*rs + j*cs] addressing covers MATLAB's column-major arrays without a copy."""
    return A.ctypes.data_as(C.POINTER(C.c_double)), A.strides[0] // 8, A.strides[1] // 8


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def model_matrices(h, K=K_HOR):
    """(Lambda, A_v, A_initp, Delta) -- getPosMat.m / dmpc_soft_bound.m:92-108 / getDeltaMat.m (host, no GPU needed)."""
    L = load()
    prm = make_params(K=K, h=h)
    n = 3 * K
    Lam, Av, A0, Dl = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, 6)), np.zeros((n, n))
    if L.dmpc_model_matrices(C.byref(prm), _dp(Lam), _dp(Av), _dp(A0), _dp(Dl)) != 0:
        raise DmpcError(L.dmpc_last_error(None).decode())
    return Lam, Av, A0, Dl


def posvel_matrix(h, K):
    L = load()
    A = np.zeros((12, 3 * K))
    if L.dmpc_posvel_matrix(float(h), int(K), _dp(A)) != 0:
        raise DmpcError(L.dmpc_last_error(None).decode())
    return A


def partition(N, G, rank):
    """(lo, count, cmax) of rank's contiguous cluster (dmpc/cpp/dmpc.cpp:1600-1625; dmpc_partition of the C ABI: host arithmetic)"""
    lo, cnt, cmax = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    if load().dmpc_partition(int(N), int(G), int(rank), C.byref(lo), C.byref(cnt), C.byref(cmax)):
        raise DmpcError(load().dmpc_last_error(None).decode())
    return lo.value, cnt.value, cmax.value


class Dmpc:
    """One solver context on one HIP device (wraps dmpc_create/dmpc_destroy)."""

    def __init__(self, variant="bound", device=0, precision="f64", **kw):
        """precision: "f64" (DMPC_PREC_F64), "mixed" (DMPC_PREC_MIXED: fp32 table / scan / rows, fp64 QP), "f32factor" (DMPC_PREC_F32FACTOR: the solver's
        inverse factor stored in fp32, refined against fp64 residuals) or "low" (both);
        device: a HIP device index, DEVICE_ALL (-100: every visible GPU from this process, agents sharded over them) or
        DEVICE_CURRENT (-1: the calling thread's current device)"""
        self._L = load()
        self.prm = make_params(variant, **kw)
        self.precision = precision
        self._ctx = self._L.dmpc_create(C.byref(self.prm), int(device), PRECISIONS[precision])
        if not self._ctx:
            raise DmpcError(self._L.dmpc_last_error(None).decode())
        self.device = device

    DEVICE_ALL, DEVICE_CURRENT = -100, -1

    @property
    def n_devices(self):
        """GPUs this context drives (1 unless created with DEVICE_ALL on a multi-GPU node)"""
        return int(self._L.dmpc_group_size(self._ctx))

    @staticmethod
    def emulate_devices(n):
        """tests: DEVICE_ALL contexts created from now on run n ranks that all sit on the current GPU (0: off)"""
        if load().dmpc_debug_emulate_devices(int(n)):
            raise DmpcError("emulate_devices: bad count")

    def debug_option(self, name, value):
        """development / tests: launch forms and tiers of this context (dmpc_debug_option: no_cull, no_persist, force_persist, tier1_qcap,
        crash_min, no_fast_exit, iter_cap, split_parts ...); never arithmetic"""
        self._chk(self._L.dmpc_debug_option(self._ctx, name.encode(), int(value)))
        return self

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.dmpc_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise DmpcError(self._L.dmpc_last_error(self._ctx).decode())

    def set_params(self, variant=None, **kw):
        base = {f: getattr(self.prm, f) for f in ("h", "rmin", "c", "alim", "Q1", "S1", "term", "max_tries")}
        base["pmin"], base["pmax"] = tuple(self.prm.pmin), tuple(self.prm.pmax)
        base.update(kw)
        self.prm = make_params(self.prm.variant if variant is None else variant, **base)
        self._chk(self._L.dmpc_set_params(self._ctx, C.byref(self.prm)))

    # ---- host-array entry points -------------------------------------------------------------
    def init_batch(self, po, pf):
        po, pf = _f(po), _f(pf)
        shp = po.shape[:-1]
        S, N = (1, shp[0]) if len(shp) == 1 else shp
        l = np.zeros(shp + (45,))
        v = np.zeros_like(l)
        a = np.zeros_like(l)
        self._chk(self._L.dmpc_init_batch(self._ctx, S, N, _dp(po), _dp(pf), _dp(l), _dp(v), _dp(a)))
        return l, v, a

    def step_batch(self, l, x_p, x_v, x_a, pf):
        """l: [N,45] or [S,N,45]; states/goals [..,3]. Returns dict(p,v,a,status,info)."""
        l, x_p, x_v, x_a, pf = _f(l), _f(x_p), _f(x_v), _f(x_a), _f(pf)
        lead = l.shape[:-1]
        S, N = (1, lead[0]) if len(lead) == 1 else lead
        p, v, a = np.zeros(lead + (45,)), np.zeros(lead + (45,)), np.zeros(lead + (45,))
        status = np.zeros(lead, dtype=np.int32)
        info = np.zeros(lead + (INFO_LEN,), dtype=np.int32)
        self._chk(self._L.dmpc_step_batch(self._ctx, S, N, _dp(l), _dp(x_p), _dp(x_v), _dp(x_a), _dp(pf), _dp(p), _dp(v),
                                          _dp(a), _ip(status), _ip(info)))
        return dict(p=p, v=v, a=a, status=status, info=info)

    def solve_one(self, l, n, po, vo, ao, pf):
        l, po, vo, ao, pf = _f(l), _f(po), _f(vo), _f(ao), _f(pf)
        p, v, a = np.zeros(45), np.zeros(45), np.zeros(45)
        status = np.zeros(1, dtype=np.int32)
        info = np.zeros(INFO_LEN, dtype=np.int32)
        self._chk(self._L.dmpc_solve_one(self._ctx, l.shape[0], int(n), _dp(l), _dp(po), _dp(vo), _dp(ao), _dp(pf), _dp(p),
                                         _dp(v), _dp(a), _ip(status), _ip(info)))
        return dict(p=p, v=v, a=a, status=int(status[0]), info=info)

    def rows_one(self, l, n, po, vo, max_rows=4096):
        """a5/a6 standalone: structured collision rows of agent n (reference order, unpruned)."""
        l, po, vo = _f(l), _f(po), _f(vo)
        xi, rhs, sc = np.zeros((max_rows, 3)), np.zeros(max_rows), np.zeros(max_rows)
        kc = np.zeros(max_rows, dtype=np.int32)
        nr, vk, st = (np.zeros(1, dtype=np.int32) for _ in range(3))
        self._chk(self._L.dmpc_rows_one(self._ctx, l.shape[0], int(n), _dp(l), _dp(po), _dp(vo), max_rows, _dp(xi), _dp(rhs),
                                        _dp(sc), _ip(kc), _ip(nr), _ip(vk), _ip(st)))
        k = min(int(nr[0]), max_rows)
        return dict(xi=xi[:k], rhs=rhs[:k], slack_coef=sc[:k], kc=kc[:k], nrows=int(nr[0]), viol_k=int(vk[0]), status=int(st[0]))

    def transition(self, po, pf, K_T_max, error_tol=0.01, histories=True):
        """histories=False: pk/vk/ak are not downloaded (they stay on the device for postcheck())."""
        po, pf = _f(po), _f(pf)
        shp = po.shape[:-1]
        S, N = (1, shp[0]) if len(shp) == 1 else shp
        used = np.zeros(S, dtype=np.int32)
        sst = np.zeros(S, dtype=np.int32)
        if histories:
            pk = np.zeros(shp + (K_T_max, 3))
            vk, ak = np.zeros_like(pk), np.zeros_like(pk)
            hp = (_dp(pk), _dp(vk), _dp(ak))
        else:
            pk = vk = ak = None
            hp = (C.POINTER(C.c_double)(),) * 3
        self._chk(self._L.dmpc_transition(self._ctx, S, N, _dp(po), _dp(pf), int(K_T_max), float(error_tol), hp[0], hp[1], hp[2],
                                          _ip(used), _ip(sst)))
        return dict(pk=pk, vk=vk, ak=ak, K_T_used=used, scene_status=sst)

    # ---- multi-GPU (one process per GPU): dmpc_multigpu.hip ----
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte RCCL id every rank passes to comm_init (hand it over by any out-of-band means)"""
        buf = C.create_string_buffer(128)
        if load().dmpc_comm_unique_id(buf):
            raise DmpcError(load().dmpc_last_error(None).decode())
        return buf.raw

    def comm_init(self, id128, nranks, rank):
        self._chk(self._L.dmpc_comm_init(self._ctx, bytes(id128), int(nranks), int(rank)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_size(self):
        """ranks of this context's communicator as RCCL counts them (ncclCommCount)"""
        return int(self._L.dmpc_comm_size(self._ctx))

    def comm_destroy(self):
        self._L.dmpc_comm_destroy(self._ctx)
        self.nranks, self.rank = 1, 0

    def step_sharded_device(self, S, N, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, lT_next, status, info, stream=0):
        self._chk(self._L.dmpc_step_sharded_device(self._ctx, S, N, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, lT_next, status, info, stream))

    def transition_sharded(self, po, pf, K_T_max, error_tol=0.01, histories=True, gather=False):
        """dmpc_transition_sharded: po, pf [S,N,3] (the same on every rank); returns this rank's histories [S,count,K_T_max,3].
        gather=True (dmpc_transition_sharded_gather): the scene-wide histories are assembled on every rank's device afterwards, so that
        postcheck(K_T_used, pf, KT_alloc=K_T_max) checks the whole transition."""
        po, pf = _f(po), _f(pf)
        S, N = po.shape[0], po.shape[1]
        lo, cnt, cmax = partition(N, getattr(self, "nranks", 1), getattr(self, "rank", 0))
        used, sst = np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int32)
        if histories:
            pk = np.zeros((S, cnt, K_T_max, 3)); vk, ak = np.zeros_like(pk), np.zeros_like(pk)
            hp = (_dp(pk), _dp(vk), _dp(ak))
        else:
            pk = vk = ak = None
            hp = (C.POINTER(C.c_double)(),) * 3
        fn = self._L.dmpc_transition_sharded_gather if gather else self._L.dmpc_transition_sharded
        self._chk(fn(self._ctx, S, N, _dp(po), _dp(pf), int(K_T_max), float(error_tol), hp[0], hp[1], hp[2], _ip(used), _ip(sst)))
        return dict(pk=pk, vk=vk, ak=ak, K_T_used=used, scene_status=sst, lo=lo, count=cnt)

    def postcheck(self, K_T_used, pf, pk=None, vk=None, ak=None, KT_alloc=None, vmax=2.0, amax=1.0, Ts=0.01, interp=False,
                  mask=None):
        """failure_rate.m:136-195 for S scenes.  pk/vk/ak [S,N,KT_alloc,3] (or [N,KT,3]); None: use the histories the
        last transition() left on the device (then KT_alloc = its K_T_max)."""
        pf = _f(pf)
        shp = pf.shape[:-1]
        S, N = (1, shp[0]) if len(shp) == 1 else shp
        used = np.ascontiguousarray(np.atleast_1d(K_T_used), dtype=np.int32)
        if pk is not None:
            pk, vk, ak = _f(pk), _f(vk), _f(ak)
            KT_alloc = pk.shape[-2]
        assert KT_alloc is not None
        out = dict(r_factor=np.zeros(S), h_scaled=np.zeros(S), n_samples=np.zeros(S, dtype=np.int32), min_dist=np.zeros(S),
                   violation=np.zeros(S, dtype=np.int32), totdist=np.zeros(S), traj_time=np.zeros(S))
        ns_alloc, p_i = 0, None
        if interp:   # upper bound of the sample count: r_factor is not known yet, so run once without and size from it
            pre = self.postcheck(K_T_used, pf, pk, vk, ak, KT_alloc, vmax, amax, Ts, False, mask)
            ns_alloc = max(int(pre["n_samples"].max()), 1)
            p_i = np.zeros((S, N, ns_alloc, 3))
        nul = C.POINTER(C.c_double)()
        msk = None if mask is None else np.ascontiguousarray(np.atleast_1d(mask), dtype=np.int32)
        self._chk(self._L.dmpc_postcheck(self._ctx, S, N, int(KT_alloc), _ip(used),
                                         _ip(msk) if msk is not None else C.POINTER(C.c_int32)(), _dp(pk) if pk is not None else nul,
                                         _dp(vk) if pk is not None else nul, _dp(ak) if pk is not None else nul, _dp(pf),
                                         float(vmax), float(amax), float(Ts), _dp(out["r_factor"]), _dp(out["h_scaled"]),
                                         _ip(out["n_samples"]), _dp(out["min_dist"]), _ip(out["violation"]), _dp(out["totdist"]),
                                         _dp(out["traj_time"]), _dp(p_i) if p_i is not None else nul, ns_alloc))
        if p_i is not None:
            out["p"] = p_i
        return out

    # ---- standalone small helpers (propStatedmpc.m, propState.m, is_inbounds.m, ReachedGoal.m) -----------
    def prop_state(self, A_p, A_v, a, A_initp=None, po=None, vo=None, off_p=None, off_v=None):
        A_p, A_v, a = _f(A_p), _f(A_v), _f(np.ravel(a))
        n_rows, n_cols = A_p.shape
        assert A_v.shape == A_p.shape and a.size == n_cols
        nul = C.POINTER(C.c_double)()
        opt = lambda x: _dp(_f(np.ravel(x))) if x is not None else nul
        A0 = _f(A_initp) if A_initp is not None else None
        p, v = np.zeros(n_rows), np.zeros(n_rows)
        self._chk(self._L.dmpc_prop_state(self._ctx, n_rows, n_cols, _dp(A_p), _dp(A_v), _dp(A0) if A0 is not None else nul, opt(po), opt(vo),
                                          opt(off_p), opt(off_v), _dp(a), _dp(p), _dp(v)))
        return p, v

    def is_inbounds(self, p, pmin, pmax):
        p = _f(np.asarray(p, float).reshape(-1, 3))
        out = np.zeros(1, dtype=np.int32)
        self._chk(self._L.dmpc_is_inbounds(self._ctx, p.shape[0], _dp(p), _dp(_f(np.ravel(pmin))), _dp(_f(np.ravel(pmax))), _ip(out)))
        return bool(out[0])

    def reached_goal(self, p, pf, error_tol):
        p, pf = _f(np.asarray(p, float).reshape(-1, 3)), _f(np.asarray(pf, float).reshape(-1, 3))
        out = np.zeros(1, dtype=np.int32)
        self._chk(self._L.dmpc_reached_goal(self._ctx, p.shape[0], _dp(p), _dp(pf), float(error_tol), _ip(out)))
        return bool(out[0])

    # ---- start / goal generators (randomTest.m, randomExchange.m) on the device ---------------
    def max_deviation(self, p, prev_p):
        """maxDeviation.m: p, prev_p [K,3] (the transposes of the MATLAB 3 x K matrices)."""
        p, prev_p = _f(p), _f(prev_p)
        out = np.zeros(1)
        self._chk(self._L.dmpc_max_deviation(self._ctx, p.shape[0], _dp(p), _dp(prev_p), _dp(out)))
        return float(out[0])

    def random_test(self, S, N, pmin, pmax, rmin, c, seed):
        """S scenes of randomTest(N,pmin,pmax,rmin,E1,order=2): (po, pf) each [S,N,3]."""
        po, pf = np.zeros((S, N, 3)), np.zeros((S, N, 3))
        self._chk(self._L.dmpc_random_test(self._ctx, S, N, _dp(_f(pmin)), _dp(_f(pmax)), float(rmin), float(c), int(seed), _dp(po), _dp(pf)))
        return po, pf

    def random_exchange(self, S, N, pmin, pmax, rmin, seed):
        """S scenes of randomExchange(N,pmin,pmax,rmin): (po, pf) each [S,N,3]."""
        po, pf = np.zeros((S, N, 3)), np.zeros((S, N, 3))
        self._chk(self._L.dmpc_random_exchange(self._ctx, S, N, _dp(_f(pmin)), _dp(_f(pmax)), float(rmin), int(seed), _dp(po), _dp(pf)))
        return po, pf

    # ---- dense collision rows (CollConstr* / AddCollConstr helpers) --------------------------
    def coll_rows(self, l, sel, k_cmp, k_blk, p, a0, rmin, c, A):
        """rows for the obstacles `sel` (0-based) of l [N_obs,K,3] at horizon column k_cmp against block k_blk of A
        [a_rows, ncols] (any strides).  Returns (Ain [n_sel, ncols], bin [n_sel], dist [n_sel])."""
        l = _f(l); A = np.asarray(A, dtype=np.float64)
        N_obs, K = l.shape[0], l.shape[1]
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        n_sel, (a_rows, ncols) = len(sel), A.shape
        if A.strides[0] % 8 or A.strides[1] % 8 or min(A.strides) <= 0:
            A = np.ascontiguousarray(A)
        base, rs, cs = _strided_base(A)
        Ain = np.zeros((n_sel, ncols)); b = np.zeros(n_sel); d = np.zeros(n_sel)
        self._chk(self._L.dmpc_coll_rows(self._ctx, K, N_obs, n_sel, _ip(sel), _dp(l), int(k_cmp), int(k_blk), _dp(_f(p)), _dp(_f(a0)),
                                         float(rmin), float(c), base, a_rows, ncols, rs, cs, _dp(Ain), ncols, 1, _dp(b), _dp(d)))
        return Ain, b, d

    def rows_dense(self, xi, kc, A):
        """structured rows (xi [nr,3], kc [nr] 1-based) -> dense Ain [nr, ncols] = -(xi . A(3kc-2:3kc, :))."""
        xi, A = _f(np.asarray(xi, float).reshape(-1, 3)), np.ascontiguousarray(A, dtype=np.float64)
        kc = np.ascontiguousarray(kc, dtype=np.int32)
        nr, (a_rows, ncols) = xi.shape[0], A.shape
        Ain = np.zeros((nr, ncols))
        self._chk(self._L.dmpc_rows_dense(self._ctx, nr, _dp(xi), _ip(kc), _dp(A), a_rows, ncols, ncols, 1, _dp(Ain), ncols, 1))
        return Ain

    def add_coll_constr(self, p, po, rmin, c, A, out_order="C"):
        """cup-SCP pairwise rows: p [N,K,3], po [N,3], A [3KN, ncols] -> (Ain [K N(N-1)/2, ncols], bin).
        out_order: "C" row-major, "F" column-major (MATLAB), "S" generic strides (a padded row-major buffer)."""
        p, po = _f(p), _f(po); A = np.asarray(A, dtype=np.float64)
        N, K = p.shape[0], p.shape[1]
        assert A.shape[0] == 3 * K * N
        if A.strides[0] % 8 or A.strides[1] % 8 or min(A.strides) <= 0:
            A = np.ascontiguousarray(A)
        base, rs, cs = _strided_base(A)
        nrows, ncols = K * N * (N - 1) // 2, A.shape[1]
        if out_order == "S":
            buf = np.zeros((nrows, 2 * ncols + 3)); Ain = buf[:, 1:2 * ncols + 1:2]
        else:
            Ain = np.zeros((nrows, ncols), order=out_order)
        b = np.zeros(nrows)
        self._chk(self._L.dmpc_add_coll_constr(self._ctx, K, N, _dp(p), _dp(po), float(rmin), float(c), base, ncols, rs, cs,
                                               Ain.ctypes.data_as(C.POINTER(C.c_double)), Ain.strides[0] // 8, Ain.strides[1] // 8, _dp(b)))
        return Ain, b

    # ---- device-pointer entry points (torch tensors: pass t.data_ptr()) ------------------------
    def step_device(self, S, G, Cn, g_local, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, lT_next, status, info, stream=0):
        self._chk(self._L.dmpc_step_device(self._ctx, S, G, Cn, g_local, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out,
                                           lT_next or None, status, info or None, stream or None))

    def table_from_rows_device(self, S, G, Cn, rows, lT, stream=0):
        self._chk(self._L.dmpc_table_from_rows_device(self._ctx, S, G, Cn, rows, lT, stream or None))

    def advance_device(self, count, p_out, v_out, a_out, status, x_p, x_v, x_a, stream=0):
        self._chk(self._L.dmpc_advance_device(self._ctx, count, p_out, v_out, a_out, status, x_p, x_v, x_a, stream or None))

    def profile(self, enable=True):
        self._chk(self._L.dmpc_profile(self._ctx, 1 if enable else 0))

    def profile_read(self):
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._chk(self._L.dmpc_profile_read(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read2(self):
        """(solve-kernel avg ms, scan+order avg ms, steps) since the previous read."""
        a, b, n = C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
        self._chk(self._L.dmpc_profile_read2(self._ctx, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    @property
    def last_solve_kernel(self):
        """profiler name of the solve kernel the last MPC step launched for the bulk of its agents (dmpc_last_solve_kernel)"""
        return self._L.dmpc_last_solve_kernel(self._ctx).decode()

    @property
    def solve_count(self):
        return int(self._L.dmpc_solve_count(self._ctx))
