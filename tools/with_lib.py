"""development: run a script against another build of THIS revision of the library (A/B runs of compile-time variants).
usage: python tools/with_lib.py multiagent_planning_amd/libdmpc_hip_x.so script.py [args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiagent_planning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
