"""CPU: the development probes under tools/ and tests/dev/ are not part of the product and are not run by the suite (most need the GPU box
or a DEV_TRACE build), but they must at least stay loadable: every Python file byte-compiles, every shell script parses."""
import glob
import os
import py_compile
import subprocess

from helpers import ROOT


def test_probe_scripts_compile():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "dev", "*.py")))
    assert len(files) > 25
    for f in files:
        py_compile.compile(f, doraise=True)


def test_probe_shell_scripts_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tests", "dev", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)


def test_key_fit_replays_a_queue(tmp_path):
    """tools/key_fit.py (the CPU half of the launch-order key's fit): on a synthetic record the perfect order must beat a random one, and the
    script must run end to end on the file format tools/gpu_key_features.py writes"""
    import numpy as np
    rng = np.random.default_rng(0)
    steps, N = 4, 3000
    info = np.zeros((steps, N, 8), dtype=np.int32)
    for k in range(steps):
        work = (rng.gamma(2.0, 80.0, N)).astype(np.int32) + 8          # quarter microseconds
        rows = np.clip((work / 40 + rng.normal(0, 3, N)).astype(np.int32), 0, 60)
        nsat, nrv, qs, ls = rng.integers(0, 45, N), np.clip(rows // 4, 0, 63), rng.integers(0, 32, N), (work > 600).astype(np.int32)
        info[k, :, 0] = rng.integers(1, 15, N); info[k, :, 1] = rows; info[k, :, 2] = 1 + ls; info[k, :, 3] = work
        info[k, :, 4] = work // 10; info[k, :, 7] = np.clip(work // 12, 0, 56)
        info[k, :, 5] = np.clip(rows, 0, 255) | (qs << 9) | (nsat << 14) | (nrv << 20) | (ls << 26)
    f = tmp_path / "key_features.npz"
    np.savez_compressed(f, info=info)
    out = subprocess.run(["python", os.path.join(ROOT, "tools", "key_fit.py"), str(f), "64"], check=True, capture_output=True, text=True).stdout
    val = {ln[:60].strip(): float(ln.split("mean makespan")[1].split("us")[0]) for ln in out.splitlines() if "mean makespan" in ln and "us" in ln.split("mean makespan")[1][:14]}
    assert val["perfect (by the work estimate itself)"] < val["random"]
    assert "best linear combination" in out
