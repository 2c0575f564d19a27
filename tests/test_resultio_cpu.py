"""CPU: the reference's result-file formats (dmpc_trajectories2file / dmpc_test2file, host code of libdmpc_hip.so).
Known answers are the reference's own result files: the header + po + pf lines of
`dmpc/cpp_results/trajectories (200-agents).txt` and the whole `cluster_test.txt` (tests/golden/)."""
import os

import numpy as np
import pytest

from multiagent_planning_amd import resultio
from helpers import GOLD

REF_DUMP = "/root/reference/dmpc/cpp_results/trajectories (200-agents).txt"


def test_trajectories_head_bytes_match_reference_file(tmp_path):
    head = open(os.path.join(GOLD, "trajectories_200_head.txt"), "rb").read()
    lines = head.decode().splitlines()
    hdr = lines[0].split()
    N, N_cmd, hs = int(hdr[0]), int(hdr[1]), float(hdr[2])
    pmin, pmax = np.array(hdr[3:6], float), np.array(hdr[6:9], float)
    po = np.array([l.split() for l in lines[1:4]], float).T
    pf = np.array([l.split() for l in lines[4:7]], float).T
    assert (N, N_cmd, hs) == (200, 200, 0.2)
    z = po[:, None, :].copy()                                   # one-column trajectories: T = 1
    out = tmp_path / "t.txt"
    resultio.write_trajectories(out, po, pf, z, z, z, hs, pmin, pmax)
    got = open(out, "rb").read()
    assert got.startswith(head)                                  # Eigen's aligned 6-digit stream format, byte for byte
    r = resultio.read_trajectories(out)
    assert np.array_equal(r["po"], po) and np.array_equal(r["pf"], pf) and np.array_equal(r["pk"][:, 0], po)
    assert np.array_equal(r["pmin"], pmin) and np.array_equal(r["pmax"], pmax) and r["h_scaled"] == hs


def test_cluster_test_file_roundtrip_is_byte_identical(tmp_path):
    src = os.path.join(GOLD, "cluster_test.txt")
    r = resultio.read_cluster_test(src)
    assert list(r["cluster_size"]) == [1, 2, 4, 6, 8, 10] and list(r["num_vehicles"]) == [10, 20] and r["times"].shape == (6, 2, 5)
    out = tmp_path / "c.txt"
    resultio.write_cluster_test(out, r["cluster_size"], r["num_vehicles"], r["times"])
    assert open(out, "rb").read() == open(src, "rb").read()


def test_trajectories_roundtrip_random(tmp_path):
    rng = np.random.default_rng(0)
    N, T = 7, 23
    po, pf = rng.uniform(-2, 2, (N, 3)), rng.uniform(-2, 2, (N, 3))
    pos, vel, acc = (rng.normal(0, s, (N, T, 3)) for s in (2.0, 1e-3, 50.0))
    out = tmp_path / "r.txt"
    resultio.write_trajectories(out, po, pf, pos, vel, acc, 0.2632148, (-2.5, -2.5, 0.2), (2.5, 2.5, 2.2))
    r = resultio.read_trajectories(out)
    for a, b in ((r["po"], po), (r["pf"], pf), (r["pk"], pos), (r["vk"], vel), (r["ak"], acc)):
        assert a.shape == b.shape and np.all(np.abs(a - b) <= 5.1e-6 * np.abs(b) + 1e-300)   # 6 significant digits
    assert abs(r["h_scaled"] - 0.263215) < 1e-12
    with pytest.raises(RuntimeError):
        resultio.write_trajectories(tmp_path / "no_such_dir" / "x.txt", po, pf, pos, vel, acc, 0.2, (0, 0, 0), (1, 1, 1))


@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="reference checkout not present (only in the build container)")
def test_full_reference_dump_rewrites_byte_identically(tmp_path):
    r = resultio.read_trajectories(REF_DUMP)
    assert r["pk"].shape == (200, 83, 3)
    out = tmp_path / "full.txt"
    resultio.write_trajectories(out, r["po"], r["pf"], r["pk"], r["vk"], r["ak"], r["h_scaled"], r["pmin"], r["pmax"])
    assert open(out, "rb").read() == open(REF_DUMP, "rb").read()
