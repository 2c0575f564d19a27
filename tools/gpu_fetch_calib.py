"""development (round 4): calibration of rocprofv3's FETCH_SIZE on THIS library's access pattern.  Streams a buffer twice the size of the
Infinity Cache (512 MiB) with coalesced 8-byte-per-lane and 16-byte-per-lane loads (dmpc_debug_read_probe); run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out> -o calib -- python tools/gpu_fetch_calib.py
and divide the known byte count by the counter (tools/profile_summary.py reads <out>/calib_counter_collection.csv)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiagent_planning_amd as mp
from multiagent_planning_amd import _lib, workload as wl
d = mp.Dmpc("bound", **wl.solver_kwargs(wl.CONFIGS["C4"], 100))
L = _lib.load()
L.dmpc_debug_read_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
BYTES = 512 << 20
for lane_bytes in (8, 16):
    assert L.dmpc_debug_read_probe(d._ctx, BYTES, lane_bytes, 3) == 0
print("read", BYTES, "bytes x 3 launches at 8 and at 16 bytes per lane")
