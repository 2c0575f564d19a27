"""development: run another tools/ script against the DEV_TRACE build of the library (multiagent_planning_amd/libdmpc_hip_trace.so,
built by `make -C multiagent_planning_amd/csrc trace`).   usage: python tools/with_trace_lib.py tools/gpu_bound_times.py [args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiagent_planning_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "multiagent_planning_amd", "libdmpc_hip_trace.so")
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
