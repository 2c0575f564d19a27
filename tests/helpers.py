"""Shared test helpers: golden loading, parameter builders for oracle and product, scenes."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

ALL_VARIANTS = ["bound", "bound2", "all3", "hard", "ondemand", "ellip", "softall", "repair", "cpp", "cpp2", "cpp1", "softall_c", "scp"]
# (scp without a `tol` keyword runs with make_params' default tol = 2, the value of dmpc/matlab/dmpc.m:14; tests/test_gpu_scp.py sweeps smaller ones)
CAMPAIGN_VARIANTS = ALL_VARIANTS[:11]   # the variant list of rounds 2-5: the recorded scenes of the randomized campaigns are positions in ITS random stream


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    kw = dict(rmin=float(g["rmin"]), c=float(g["c"]), alim=float(g["alim"]), Q1=float(g["Q"]), S1=float(g["S"]),
              term=float(g["term"]), pmin=tuple(g["pmin"]), pmax=tuple(g["pmax"]), h=float(g["h"]))
    return g, kw


def oracle_params(variant, kw):
    from oracle import oracle as orc
    return orc.make_params(variant, **kw)


def init_table(po, pf, h=0.2, K=15):
    """initDMPC.m straight-line predictions for all agents, rows [N,45]."""
    t = np.arange(K) * h
    return (po[:, None, :] + t[None, :, None] * (pf - po)[:, None, :] / 10).reshape(po.shape[0], 3 * K)


def step14_inputs(g):
    return g["l"], g["pk"][:, 12], g["vk"][:, 12], g["ak"][:, 12], g["pf"]


def compare_to_oracle(out, ref, tol=1e-9, what=""):
    """Teacher-forced parity of one MPC step: identical branch records, trajectory l_inf <= tol."""
    st_o, st_r = np.asarray(out["status"]).ravel(), np.asarray(ref["status"]).ravel()
    assert np.array_equal(st_o, st_r), f"{what}: status mismatch at {np.where(st_o != st_r)[0][:10]}: {st_o[st_o != st_r][:10]} vs {st_r[st_o != st_r][:10]}"
    io, ir = np.asarray(out["info"]).reshape(-1, 8), np.asarray(ref["info"]).reshape(-1, 8)
    # product info: violk nrows tries case ... ; oracle info: violk nv tries case iters nslack nact nrows
    assert np.array_equal(io[:, 0], ir[:, 0]), f"{what}: viol_k mismatch"
    assert np.array_equal(io[:, 1], ir[:, 7]), f"{what}: row count mismatch"
    assert np.array_equal(io[:, 3], ir[:, 3]), f"{what}: cost case mismatch"
    assert np.array_equal(io[:, 2], ir[:, 2]), f"{what}: retry-ladder tries mismatch {io[:,2][io[:,2]!=ir[:,2]]} vs {ir[:,2][io[:,2]!=ir[:,2]]}"
    solved = (st_r & 1) == 1
    errs = {}
    for key in ("p", "v", "a"):
        o, r = np.asarray(out[key]).reshape(-1, 45), np.asarray(ref[key]).reshape(-1, 45)
        e = np.abs(o[solved] - r[solved]).max() if solved.any() else 0.0
        errs[key] = e
        assert e <= tol, f"{what}: l_inf({key}) = {e:.3e} > {tol:.1e}"
        assert np.all(o[~solved] == 0.0), f"{what}: outputs of unsolved agents must be zero"
    return errs


def unrescale(g):
    """Recover the un-rescaled MPC histories behind a recorded post-check block: a_k = recorded/r_factor (the last
    column is never rescaled, failure_rate.m:156), and v, p follow the MPC model x_k = A x_{k-1} + b a_k the
    histories were produced with (the rescale loop only needs a_k, p_1, v_1; v_k enters r_factor)."""
    rf, h = float(g["r_factor"]), float(g["h"])
    a = g["ak"] / rf
    a[:, -1] = g["ak"][:, -1]
    p, v = np.zeros_like(a), np.zeros_like(a)
    p[:, 0], v[:, 0] = g["pk"][:, 0], g["vk"][:, 0]
    for k in range(1, a.shape[1]):
        v[:, k] = v[:, k - 1] + h * a[:, k]
        p[:, k] = p[:, k - 1] + h * v[:, k - 1] + h * h / 2 * a[:, k]
    return p, v, a
