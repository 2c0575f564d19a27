/* dmpc_hip_dev.h -- DEVELOPMENT interface of libdmpc_hip.so: what the test suite and the probes under tools/ use to select launch forms,
 * emulate ranks and read internal records.  NOT part of the drop-in boundary (include/dmpc_hip.h): nothing here replaces a reference
 * interface, none of it changes a result beyond solver round-off (most not a bit: tests/test_gpu_paths.py), and it may change between
 * rounds.  Exported so that the checks run against the PRODUCT binary instead of a special build. */
#ifndef DMPC_HIP_DEV_H
#define DMPC_HIP_DEV_H
#include "dmpc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* launch forms / tiers / pre-passes of a context by name (the list: INTEGRATION.md section 6); also through ONE environment variable read at
 * context creation, DMPC_DEBUG_OPTIONS="name=value,..." */
DMPC_API int dmpc_debug_option(dmpc_ctx *ctx, const char *name, int value);
/* DMPC_DEVICE_ALL contexts created from now on run n ranks that all sit on the calling thread's current device (0: off) -- the
 * single-process multi-GPU protocol on a one-GPU box */
DMPC_API int dmpc_debug_emulate_devices(int n);
/* a context without a communicator acts as rank `rank` of `nranks` (the ranks of a job run one after the other on one GPU) */
DMPC_API int dmpc_debug_set_rank(dmpc_ctx *ctx, int nranks, int rank);
/* DEV_TRACE builds: per-iteration trace of one agent / per-wave and per-agent clocks (agent = -2, -3, -5) / crash statistics (-4) */
DMPC_API int dmpc_debug_trace(dmpc_ctx *ctx, int agent, int cap, double *host_out);
/* the scan's hand-off headers of the last step (8 ints per agent) */
DMPC_API int dmpc_debug_read_hdr(dmpc_ctx *ctx, int *host_out, int n_agents);
/* force the solve launch order of the next launches (n = 0: the built-in policy) */
DMPC_API int dmpc_debug_set_order(dmpc_ctx *ctx, const int *host_order, int n);
/* a coalesced streaming read of `bytes` bytes at 8 or 16 bytes per lane, `reps` launches: the known byte count rocprofv3's FETCH_SIZE is
 * calibrated on (tools/gpu_fetch_calib.py) */
DMPC_API int dmpc_debug_read_probe(dmpc_ctx *ctx, size_t bytes, int lane_bytes, int reps);
#ifdef __cplusplus
}
#endif
#endif
