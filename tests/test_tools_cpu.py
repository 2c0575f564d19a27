"""CPU: the development probes under tools/ and tests/dev/ are not part of the product and are not run by the suite (most need the GPU box
or a DEV_TRACE build), but they must at least stay loadable: every Python file byte-compiles, every shell script parses."""
import glob
import os
import py_compile
import subprocess

from helpers import ROOT


def test_probe_scripts_compile():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "dev", "*.py")))
    assert len(files) > 40
    for f in files:
        py_compile.compile(f, doraise=True)


def test_probe_shell_scripts_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tests", "dev", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)
