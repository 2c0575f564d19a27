"""The MATLAB side of the boundary without MATLAB: the MEX gateway (multiagent_planning_amd/matlab/dmpc_mex.cpp) compiled
against the mock MEX runtime of tests/mock_mex/ and driven the way MATLAB would call it (column-major mxArrays, a params
struct, a command string).  CPU: it compiles, links the C ABI and runs the host-only commands; GPU: every command against
the ctypes binding of the same C ABI."""
import numpy as np
import pytest

import mexharness as mh
from helpers import load_golden, step14_inputs

KW = dict(h=0.2, rmin=0.35, c=2.0, alim=1.0, Q1=1000.0, S1=100.0, term=-5e4, pmin=(-2.5, -2.5, 0.2), pmax=(2.5, 2.5, 2.2))


def test_gateway_compiles_and_host_commands_match_the_recorded_matlab_matrices():
    g, kw = load_golden("failure_rate2_bound")
    prm = mh.params("bound", kw)
    Lt, Avt, A0t, Dlt = mh.call("model_matrices", prm, nlhs=4)
    # the gateway returns the row-major C matrices in column-major arrays: transposes, undone by the .m wrappers
    assert np.array_equal(Lt.T, g["A"]) and np.array_equal(Avt.T, g["A_v"])
    assert np.array_equal(A0t.T, g["A_initp"]) and np.array_equal(Dlt.T, g["Delta"])
    Aaug, = mh.call("posvel_matrix", prm)
    assert Aaug.shape == (12, 45)
    assert np.allclose(Aaug[:3], g["A"][-3:]) and np.allclose(Aaug[3:6], g["A_v"][-3:])      # final position / velocity rows
    assert np.array_equal(Aaug[6:9, -3:], np.eye(3)) and np.array_equal(Aaug[9:12, :3], np.eye(3))


def test_gateway_reports_errors_like_matlab():
    # an unknown command raises dmpc:cmd on a GPU box; without a HIP device the context creation fails first and LOUDLY
    # (dmpc:create, "no CPU fallback") -- either way a MATLAB error with an identifier, never a silent result
    with pytest.raises(RuntimeError, match="dmpc:(cmd|create)"):
        mh.call("model_matrix", mh.params("bound", KW))
    with pytest.raises(RuntimeError, match="dmpc:(shape|create)"):
        mh.call("solve_one", mh.params("bound", KW), [np.zeros((3, 15, 2)), 1, np.zeros(3)], nlhs=1)   # wrong argument count


@pytest.mark.gpu
def test_gateway_commands_match_the_c_abi():
    import multiagent_planning_amd as mp
    g, kw = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pf = step14_inputs(g)
    N = l.shape[0]
    prm = mh.params("bound2", kw)
    d = mp.Dmpc("bound2", **kw)
    lm = np.ascontiguousarray(l.reshape(N, 15, 3).transpose(2, 1, 0))          # MATLAB l(3,K,N)
    # solve_one (1-based n like the .m signature)
    for n in (0, 5, 57):
        p, v, a, st, inf = mh.call("solve_one", prm, [lm, n + 1, xp[n], xv[n], xa[n], pf[n]], nlhs=5)
        r = d.solve_one(l, n, xp[n], xv[n], xa[n], pf[n])
        assert int(st.ravel()[0]) == r["status"]
        assert np.array_equal(p.T.ravel(), r["p"]) and np.array_equal(a.T.ravel(), r["a"]) and np.array_equal(v.T.ravel(), r["v"])
    # step_batch
    P, V, A, st, inf = mh.call("step_batch", prm, [lm, xp.T, xv.T, xa.T, pf.T], nlhs=5)
    out = d.step_batch(l, xp, xv, xa, pf)
    assert np.array_equal(st.ravel(), out["status"])
    assert np.array_equal(P.transpose(2, 1, 0).reshape(N, 45), out["p"]) and np.array_equal(A.transpose(2, 1, 0).reshape(N, 45), out["a"])
    # init_batch == initDMPC.m
    p0, v0, a0 = mh.call("init_batch", prm, [g["po"].T, g["pf"].T], nlhs=3)
    l0, _, _ = d.init_batch(g["po"], g["pf"])
    assert np.array_equal(p0.transpose(2, 1, 0).reshape(N, 45), l0) and not v0.any() and not a0.any()
    # rows_one + rows_dense == CheckCollSoftDMPC / CollConstrSoftDMPC2 of the variant
    viol = np.where(out["info"][:, 0] > 0)[0]
    n = int(viol[0])
    xi, rhs, dist, kc, vk, coll = mh.call("rows_one", prm, [lm, n + 1, xp[n], xv[n]], nlhs=6)
    rr = d.rows_one(l, n, xp[n], xv[n])
    assert int(vk.ravel()[0]) == rr["viol_k"] and np.array_equal(xi.T, rr["xi"]) and np.array_equal(rhs.ravel(), rr["rhs"])
    Lam, Av, A0, Dl = mp.model_matrices(kw["h"])
    Ain, = mh.call("rows_dense", prm, [xi, kc, Lam])
    assert np.array_equal(Ain, d.rows_dense(rr["xi"], rr["kc"], Lam))
    # prop_state == propStatedmpc.m
    ok = int(np.where(out["status"] == 1)[0][0])
    pp, vv = mh.call("prop_state", prm, [Lam, Av, A0, xp[ok], xv[ok], out["a"][ok]], nlhs=2)
    assert np.abs(pp.ravel() - out["p"][ok]).max() < 1e-12 and np.abs(vv.ravel() - out["v"][ok]).max() < 1e-12
    # is_inbounds.m / ReachedGoal.m (what the .m shims of the same names call)
    pts = out["p"][ok].reshape(15, 3)
    ib, = mh.call("is_inbounds", prm, [pts.T, np.array(kw["pmin"]), np.array(kw["pmax"])])
    assert bool(ib.ravel()[0]) == bool(d.is_inbounds(pts, kw["pmin"], kw["pmax"]))
    ib2, = mh.call("is_inbounds", prm, [(pts + 100.0).T, np.array(kw["pmin"]), np.array(kw["pmax"])])
    assert not bool(ib2.ravel()[0])
    rg, = mh.call("reached_goal", prm, [xp.T, pf.T, 0.05])
    assert bool(rg.ravel()[0]) == bool(d.reached_goal(xp, pf, 0.05))
    rg2, = mh.call("reached_goal", prm, [pf.T, pf.T, 0.05])
    assert bool(rg2.ravel()[0])
    # solveDMPC.m / solveSoftDMPC_c.m shims: 'solve_one' on a context of variant 12 (params.tol = the .m's tol) / 11; maxDeviation.m: 'max_deviation'
    dscp, dsc = mp.Dmpc("scp", tol=0.05, **kw), mp.Dmpc("softall_c", **kw)
    for n in (0, 5, 57):
        for pr, dd in ((mh.params("scp", kw, tol=0.05), dscp), (mh.params("softall_c", kw), dsc)):
            p, v, a, st, inf = mh.call("solve_one", pr, [lm, n + 1, xp[n], xv[n], xa[n], pf[n]], nlhs=5)
            r = dd.solve_one(l, n, xp[n], xv[n], xa[n], pf[n])
            assert int(st.ravel()[0]) == r["status"] and np.array_equal(inf.ravel(), r["info"])
            assert np.array_equal(p.T.ravel(), r["p"]) and np.array_equal(a.T.ravel(), r["a"]) and np.array_equal(v.T.ravel(), r["v"])
    pa = out["p"][ok].reshape(15, 3).T
    md, = mh.call("max_deviation", prm, [pa, lm[:, :, ok]])
    assert md.ravel()[0] == max(np.linalg.norm(pa[:, k] - lm[:, k, ok]) for k in range(5))
    # generators and a whole transition (C1: the 4-agent swap of dmpc_soft_bound.m)
    from multiagent_planning_amd import workload as wl
    c1 = wl.CONFIGS["C1"]
    kw1 = wl.solver_kwargs(c1)
    prm1 = mh.params("bound", kw1)
    po, pfs = np.asarray(c1["po"], float), np.asarray(c1["pf"], float)
    pk, vk_, ak, used, sst = mh.call("transition", prm1, [po.T, pfs.T, c1["K_T"], c1["error_tol"]], nlhs=5)
    ref = mp.Dmpc("bound", **kw1).transition(po[None], pfs[None], c1["K_T"], c1["error_tol"])
    assert int(used.ravel()[0]) == int(ref["K_T_used"][0]) and int(sst.ravel()[0]) == int(ref["scene_status"][0])
    assert np.array_equal(pk.transpose(2, 1, 0), ref["pk"][0])
    rpo, rpf = mh.call("random_test", prm1, [20, kw1["pmin"], kw1["pmax"], 0.5, 1.5, 42], nlhs=2)
    qpo, qpf = mp.Dmpc("bound", **kw1).random_test(1, 20, kw1["pmin"], kw1["pmax"], 0.5, 1.5, 42)
    assert np.array_equal(rpo.T, qpo[0]) and np.array_equal(rpf.T, qpf[0])


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 3])
def test_gateway_transition_and_step_on_all_gpus_of_one_process(G):
    """The gateway creates its context on EVERY visible GPU (DMPC_DEVICE_ALL): the agents of a scene are sharded over the GPUs inside
    the library, one host thread per GPU and peer copies between MPC steps (the thread clusters of dmpc.cpp:1600-1625,1656-1686).  On
    this one-GPU box the ranks are emulated on the same device: dmpc_mex('transition') and dmpc_mex('step_batch') must return what
    the single-GPU context returns, bit for bit (G = 3: unequal clusters)."""
    import multiagent_planning_amd as mp
    from multiagent_planning_amd import workload as wl
    cfg = wl.CONFIGS["C4"]
    N = 10
    kw = wl.solver_kwargs(cfg, N)
    prm = mh.params("bound", kw)
    po, pf = wl.make_scenes(cfg, 1, N, wl.SEED0 + 5)
    pk, vk, ak, used, sst = mh.call("transition", prm, [po[0].T, pf[0].T, 60, cfg["error_tol"]], nlhs=5, emulate_devices=G)
    ref = mp.Dmpc("bound", **kw).transition(po, pf, 60, cfg["error_tol"])
    assert int(used.ravel()[0]) == int(ref["K_T_used"][0]) and int(sst.ravel()[0]) == int(ref["scene_status"][0])
    assert int(sst.ravel()[0]) & mp.ST_REACHED
    for got, want in ((pk, "pk"), (vk, "vk"), (ak, "ak")):
        assert np.array_equal(got.transpose(2, 1, 0), ref[want][0]), want
    g, kwg = load_golden("comp_kctr_3_bound2")
    l, xp, xv, xa, pfg = step14_inputs(g)
    Ng = l.shape[0]
    lm = np.ascontiguousarray(l.reshape(Ng, 15, 3).transpose(2, 1, 0))
    P, V, A, st, inf = mh.call("step_batch", mh.params("bound2", kwg), [lm, xp.T, xv.T, xa.T, pfg.T], nlhs=5, emulate_devices=G)
    out = mp.Dmpc("bound2", **kwg).step_batch(l, xp, xv, xa, pfg)
    assert np.array_equal(st.ravel(), out["status"])
    assert np.array_equal(P.transpose(2, 1, 0).reshape(Ng, 45), out["p"]) and np.array_equal(A.transpose(2, 1, 0).reshape(Ng, 45), out["a"])
    assert np.array_equal(V.transpose(2, 1, 0).reshape(Ng, 45), out["v"])
