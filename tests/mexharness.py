"""Test helper: compiles the MEX gateway (multiagent_planning_amd/matlab/dmpc_mex.cpp) against the mock MEX runtime of
tests/mock_mex/ and calls its mexFunction() through tests/mock_mex/harness.cpp -- the MATLAB side of the boundary,
exercised without MATLAB."""
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_mex")
EXE = os.path.join(MOCK, "_build", "mex_harness")
SRC = [os.path.join(MOCK, "harness.cpp"), os.path.join(MOCK, "mock_mex.cpp"),
       os.path.join(ROOT, "multiagent_planning_amd", "matlab", "dmpc_mex.cpp")]


def build():
    libdir = os.path.join(ROOT, "multiagent_planning_amd")
    deps = SRC + [os.path.join(MOCK, "mex.h"), os.path.join(ROOT, "include", "dmpc_hip.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(p) for p in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-misleading-indentation", "-I", MOCK, "-I", os.path.join(ROOT, "include"),
                               "-o", EXE] + SRC + ["-L", libdir, "-ldmpc_hip", "-Wl,-rpath," + libdir])
    return EXE


def call(cmd, prm, args=(), nlhs=1, emulate_devices=0):
    """dmpc_mex(cmd, prm, args...) -> list of nlhs numpy arrays (column-major like MATLAB); raises RuntimeError with the
    MATLAB error text on mexErrMsgIdAndTxt.  prm: dict with K, variant, order, h, rmin, c, alim, Q1, S1, term, pmin, pmax."""
    exe = build()
    req = struct.pack("<i", len(cmd)) + cmd.encode()
    req += struct.pack("<3i", int(prm["K"]), int(prm["variant"]), int(prm["order"]))
    req += struct.pack("<14d", prm["h"], prm["rmin"], prm["c"], prm["alim"], prm["Q1"], prm["S1"], prm["term"], *prm["pmin"], *prm["pmax"], prm.get("tol", 0.0))
    req += struct.pack("<i", len(args))
    for a in args:
        a = np.asarray(a, dtype=np.float64)
        if a.ndim < 2:
            a = a.reshape(1, -1) if a.ndim == 1 else a.reshape(1, 1)
        req += struct.pack("<i", a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape) + a.tobytes(order="F")
    req += struct.pack("<i", nlhs)
    with tempfile.TemporaryDirectory() as td:
        rq, rp = os.path.join(td, "req.bin"), os.path.join(td, "rep.bin")
        open(rq, "wb").write(req)
        env = dict(os.environ)
        if emulate_devices:
            env["DMPC_TEST_EMULATE_DEVICES"] = str(int(emulate_devices))
        subprocess.check_call([exe, rq, rp], env=env)
        buf = open(rp, "rb").read()
    rc, = struct.unpack_from("<i", buf, 0)
    off = 4
    if rc:
        n, = struct.unpack_from("<i", buf, off)
        raise RuntimeError(buf[off + 4: off + 4 + n].decode())
    nout, = struct.unpack_from("<i", buf, off); off += 4
    outs = []
    for _ in range(nout):
        cls, nd = struct.unpack_from("<2i", buf, off); off += 8
        dims = struct.unpack_from(f"<{nd}q", buf, off); off += 8 * nd
        dt = np.int32 if cls == 12 else np.float64
        cnt = int(np.prod(dims))
        outs.append(np.frombuffer(buf, dtype=dt, count=cnt, offset=off).reshape(dims, order="F").copy())
        off += cnt * np.dtype(dt).itemsize
    return outs


def params(variant, kw, K=15, order=2, tol=0.0):
    from multiagent_planning_amd import _lib
    return dict(K=K, variant=_lib.VARIANTS[variant], order=order, h=kw["h"], rmin=kw["rmin"], c=kw["c"], alim=kw["alim"], Q1=kw["Q1"],
                S1=kw["S1"], term=kw["term"], pmin=tuple(kw["pmin"]), pmax=tuple(kw["pmax"]), tol=tol)
