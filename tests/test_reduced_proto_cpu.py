"""The reduced dual active-set method of csrc/dmpc_rsolve.hip on the CPU: its prototype (tools/proto/rqp_proto.c -- bounds and slack bounds as fixed
variables, soft rows as a 3x3 penalty, every equality-constrained QP from scratch) against the oracle on a scene at the headline's density, with the
kernel's pivot rule and with random pivot orders (every path of the method must end at the same minimiser, with the same retry-ladder count)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _proto():
    spec = importlib.util.spec_from_file_location("run_proto", os.path.join(ROOT, "tools", "proto", "run_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_reduced_method_prototype_matches_the_oracle():
    m = _proto()
    for jitter in (False, True):
        tot = m.run(300, 4, 20180926 + 61, "bound", jitter=jitter, quiet=True)
        assert tot["n"] >= 1100
        assert tot["mism_status"] == 0 and tot["mism_tries"] == 0 and tot["fallback"] == 0, tot
        assert tot["maxerr"] <= 1e-9, tot
