"""The reference's on-disk result formats.

  write_trajectories / read_trajectories   DMPC::trajectories2file (dmpc/cpp/dmpc.cpp:2088-2126) and its reader
                                           dmpc/cpp_results/read_result.m:4-44
  write_cluster_test / read_cluster_test   test2file (dmpc/cpp/cluster_test.cpp:9-33) and dmpc/cpp_results/cluster_test.m

The writers are the C functions of libdmpc_hip.so (byte-compatible with Eigen's stream format); the readers mirror the
MATLAB scripts (`dlmread(...,'')` pads short rows with zeros; so do these).
"""
import ctypes as C

import numpy as np

from . import _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def write_trajectories(path, po, pf, pos, vel, acc, h_scaled, pmin, pmax):
    """po [N,3], pf [N_cmd,3], pos/vel/acc [N_cmd,T,3]."""
    L = _lib.load()
    po, pf, pos, vel, acc = _f(po), _f(pf), _f(pos), _f(vel), _f(acc)
    N, (N_cmd, T, _) = po.shape[0], pos.shape
    assert pf.shape == (N_cmd, 3) and vel.shape == pos.shape and acc.shape == pos.shape
    if L.dmpc_trajectories2file(str(path).encode(), N, N_cmd, T, float(h_scaled), _dp(_f(pmin)), _dp(_f(pmax)), _dp(po), _dp(pf),
                                _dp(pos), _dp(vel), _dp(acc)):
        raise _lib.DmpcError(L.dmpc_last_error(None).decode())


def _dlmread(path):
    rows = [np.array(line.split(), dtype=np.float64) for line in open(path) if line.strip()]
    M = np.zeros((len(rows), max(len(r) for r in rows)))
    for i, r in enumerate(rows):
        M[i, :len(r)] = r
    return M


def read_trajectories(path):
    """read_result.m:4-44 -> dict(N, N_cmd, h_scaled, pmin, pmax, po [N,3], pf [N_cmd,3], pk/vk/ak [N_cmd,T,3])."""
    M = _dlmread(path)
    N, N_cmd = int(M[0, 0]), int(M[0, 1])
    out = dict(N=N, N_cmd=N_cmd, h_scaled=M[0, 2], pmin=M[0, 3:6].copy(), pmax=M[0, 6:9].copy(),
               po=M[1:4, :N].T.copy(), pf=M[4:7, :N_cmd].T.copy())
    T = max(len(line.split()) for i, line in enumerate(open(path)) if i >= 7)
    start = 7
    for name in ("pk", "vk", "ak"):
        blk = M[start:start + 3 * N_cmd, :T]
        out[name] = blk.reshape(N_cmd, 3, T).transpose(0, 2, 1).copy()
        start += 3 * N_cmd
    return out


def write_cluster_test(path, cluster_size, num_vehicles, times):
    """times [n_cluster, n_vehicles, n_trials] wall times."""
    L = _lib.load()
    cs, nv, t = _f(cluster_size), _f(num_vehicles), _f(times)
    assert t.shape[:2] == (cs.size, nv.size)
    if L.dmpc_test2file(str(path).encode(), cs.size, nv.size, t.shape[2], _dp(cs), _dp(nv), _dp(t)):
        raise _lib.DmpcError(L.dmpc_last_error(None).decode())


def read_cluster_test(path):
    M = _dlmread(path)
    nc, nv, nt = (int(x) for x in M[0, :3])
    return dict(cluster_size=M[1, :nc].copy(), num_vehicles=M[1, nc:nc + nv].copy(),
                times=M[2:2 + nc * nv, :nt].reshape(nc, nv, nt).copy())
