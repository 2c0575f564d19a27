// dmpc_multigpu.hip -- agents of every scene sharded over the GPUs of one node, one process (rank) per GPU, inside the
// library (included by dmpc_api.hip).
//
// The reference shards the agents of a transition over host threads in contiguous clusters -- N/G each, the first N mod G
// clusters one more (DMPC::solveParallelDMPCv2, dmpc/cpp/dmpc.cpp:1600-1625) -- and every cluster reads the predictions of
// ALL agents from the previous MPC step (`prev_obs = obs` after the join, :1671-1681; `l = new_l`, dmpc_soft_bound.m:146).
// Here a cluster is a rank: the table lT[G][S][3K][Cmax] is rank-major (Cmax = ceil(N/G); the last column of the short
// ranks' chunks is padding the kernels never read, StepParams::short_from), every rank solves its own chunk against the whole
// table, and ONE RCCL all-gather over xGMI per MPC step (S*3K*Cmax doubles per rank, issued on the context's stream right
// behind the solve) writes every rank's new predictions straight into its slot of the next table.  The termination test
// (ReachedGoal.m / the abort of failure_rate.m:112-125) needs one (reached, status-OR) pair per scene and rank: a second,
// tiny all-gather grouped with the first.  RCCL is loaded with dlopen when a communicator is first asked for, so the
// single-GPU library has no link-time dependency on it (and binds to the copy PyTorch already loaded, if any).
#include <dlfcn.h>
#include <mutex>

namespace mg {

typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;
enum { kDouble = 8, kFloat = 7, kInt32 = 2 };      // ncclDataType_t: ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8 (rccl.h)

struct Api {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*CommCount)(Comm, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

static void api_load(Api &a);
static Api *api()
{
    static Api a;
    static std::once_flag once;   // contexts on several host threads may ask at the same time
    std::call_once(once, [&]() { api_load(a); });
    return &a;
}
static void api_load(Api &a)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) { a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.h) break; }
    if (!a.h) { a.err = std::string("cannot load librccl: ") + dlerror(); return; }
    auto sym = [&](const char *n) -> void * { void *p = dlsym(a.h, n); if (!p && a.err.empty()) a.err = std::string("librccl lacks ") + n; return p; };
    a.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
    a.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
    a.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    a.CommAbort = (int (*)(Comm))dlsym(a.h, "ncclCommAbort");   // (optional: a failing rank releases its peers)
    a.CommCount = (int (*)(Comm, int *))dlsym(a.h, "ncclCommCount");   // (optional: dmpc_comm_size reports what the communicator itself says)
    a.AllGather = (int (*)(const void *, void *, size_t, int, Comm, hipStream_t))sym("ncclAllGather");
    a.GroupStart = (int (*)())sym("ncclGroupStart");
    a.GroupEnd = (int (*)())sym("ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
}

// rows [S][N][3K] -> lT[G][S][3K][Cmax] for the contiguous clusters of dmpc.cpp:1600-1625 (rem = N mod G clusters of Cmax
// agents first, then clusters of Cmax-1; padding columns are zero and never read)
__global__ void table_from_rows_padded_kernel(int S, int N, int G, int Cmax, int rem, const double *__restrict__ rows, double *__restrict__ lT)
{
    const size_t total = (size_t)G * S * N3 * Cmax;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % Cmax);
        size_t u = t / Cmax;
        const int j = (int)(u % N3); u /= N3;
        const int s = (int)(u % S);
        const int g = (int)(u / S);
        const int cnt = (rem == 0 || g < rem) ? Cmax : Cmax - 1;
        const int lo = (rem == 0) ? g * Cmax : (g < rem ? g * Cmax : rem * Cmax + (g - rem) * (Cmax - 1));
        lT[t] = c < cnt ? rows[((size_t)s * N + lo + c) * N3 + j] : 0.0;
    }
}
// own agents' slices of the full [S][N][3] arrays -> [S][cnt][3]
__global__ void slice_agents_kernel(int S, int N, int lo, int cnt, const double *__restrict__ full, double *__restrict__ own)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * cnt * 3) return;
    const int d = i % 3, a = (i / 3) % cnt, s = i / (3 * cnt);
    own[i] = full[((size_t)s * N + lo + a) * 3 + d];
}
// flags_all [G][S][2] (reached by this rank's agents, OR of their status bits) -> per-scene verdict of the step + scene_done
__global__ void combine_flags_kernel(int G, int S, const int *__restrict__ flags_all, int *__restrict__ step_flags, int *__restrict__ scene_done)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    int reached = 1, sor = 0;
    for (int g = 0; g < G; ++g) { reached &= flags_all[((size_t)g * S + s) * 2]; sor |= flags_all[((size_t)g * S + s) * 2 + 1]; }
    step_flags[(size_t)s * 2] = reached; step_flags[(size_t)s * 2 + 1] = sor;
    if (reached || (sor & ~ST_SOLVED)) scene_done[s] = 1;   // the trial of this scene is over (failure_rate.m:112-125)
}

// histories of the own agents [S][cnt][KT][3] <-> their slab of the scene-wide histories [S][N][KT][3]; padded: [S][cmax][KT][3]
__global__ void hist_place_kernel(int S, int N, int KT, int lo, int cnt, int src_agents /* cnt or cmax */, const double *__restrict__ own,
                                  double *__restrict__ full)
{
    const size_t per = (size_t)KT * 3, total = (size_t)S * cnt * per;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t % per, a = (t / per) % cnt, sc = t / (per * cnt);
        full[((size_t)sc * N + lo + a) * per + e] = own[((size_t)sc * src_agents + a) * per + e];
    }
}
__global__ void hist_pad_kernel(int S, int KT, int cnt, int cmax, const double *__restrict__ own, double *__restrict__ padded)
{
    const size_t per = (size_t)KT * 3, total = (size_t)S * cmax * per;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t % per, a = (t / per) % cmax, sc = t / (per * cmax);
        padded[t] = a < (size_t)cnt ? own[((size_t)sc * cnt + a) * per + e] : 0.0;
    }
}
__global__ void chunk_to_f32_kernel(size_t n, const double *__restrict__ src, float *__restrict__ dst)
{
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) dst[t] = (float)src[t];
}

}   // namespace mg


// contiguous clusters of dmpc.cpp:1600-1625: N/G agents each, the first N mod G one more
extern "C" int dmpc_partition(int N, int G, int rank, int32_t *lo, int32_t *count, int32_t *cmax)
{
    if (N < 1 || G < 1 || rank < 0 || rank >= G || G > N) { g_err = "dmpc_partition: need 1 <= G <= N and 0 <= rank < G"; return -1; }
    const int base = N / G, rem = N % G;
    if (count) *count = base + (rank < rem ? 1 : 0);
    if (lo) *lo = rank * base + (rank < rem ? rank : rem);
    if (cmax) *cmax = base + (rem ? 1 : 0);
    return 0;
}

extern "C" int dmpc_comm_unique_id(char *id128)
{
    mg::Api *a = mg::api();
    if (!a->err.empty()) { g_err = "dmpc_comm_unique_id: " + a->err; return -1; }
    mg::UniqueId id;
    const int rc = a->GetUniqueId(&id);
    if (rc) { g_err = std::string("ncclGetUniqueId: ") + a->GetErrorString(rc); return -1; }
    memcpy(id128, id.internal, 128);
    return 0;
}

extern "C" int dmpc_comm_init(dmpc_ctx *ctx, const char *id128, int nranks, int rank)
{
    if (!ctx) { g_err = "dmpc_comm_init: ctx is NULL"; return -1; }
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) FAIL(ctx, "dmpc_comm_init: bad arguments");
    if (ctx->comm) FAIL(ctx, "dmpc_comm_init: the context already has a communicator");
    if (ctx->grp) FAIL(ctx, "dmpc_comm_init: a DMPC_DEVICE_ALL context shards inside one process and takes no communicator");
    mg::Api *a = mg::api();
    if (!a->err.empty()) FAIL(ctx, "dmpc_comm_init: " + a->err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    mg::UniqueId id;
    memcpy(id.internal, id128, 128);
    mg::Comm c = nullptr;
    const int rc = a->CommInitRank(&c, nranks, id, rank);
    if (rc) FAIL(ctx, std::string("ncclCommInitRank: ") + a->GetErrorString(rc));
    ctx->comm = c; ctx->nranks = nranks; ctx->rank = rank;
    return 0;
}

// development aid (not part of the public header): act as rank `rank` of `nranks` WITHOUT a communicator -- the exchange then
// only fills this rank's slot, so a test can run the ranks of a sharded step one after the other on one GPU
extern "C" int dmpc_debug_set_rank(dmpc_ctx *ctx, int nranks, int rank)
{
    if (!ctx || ctx->comm || ctx->grp || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    ctx->nranks = nranks; ctx->rank = rank; ctx->debug_rank = nranks > 1;
    return 0;
}

// ranks of the context's communicator as RCCL counts them (ncclCommCount); without a communicator: the ranks of the group (DMPC_DEVICE_ALL)
// or 1.  What a launcher prints next to a scaling figure: the number of ranks that really took part.
extern "C" int dmpc_comm_size(const dmpc_ctx *ctx)
{
    if (!ctx) return 0;
    if (ctx->comm && mg::api()->CommCount) { int n = 0; if (mg::api()->CommCount((mg::Comm)ctx->comm, &n) == 0) return n; }
    return ctx->grp ? ctx->grp->G : ctx->nranks;
}

extern "C" int dmpc_comm_destroy(dmpc_ctx *ctx)
{
    if (!ctx) return -1;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        mg::api()->CommDestroy((mg::Comm)ctx->comm);
        ctx->comm = nullptr;
    }
    if (!ctx->grp) { ctx->nranks = 1; ctx->rank = 0; }
    ctx->debug_rank = 0;
    return 0;
}


// the exchange of one MPC step: every rank's chunk [S][3K][Cmax] (`elem` bytes per entry: 8, or 4 for the fp32 table of a mixed-
// precision transition) into its slot of the next table (+ optionally the flags).  Three transports, one protocol:
//   * no communicator, no group: a single rank (the exchange is a device copy), or -- tests, dmpc_debug_set_rank -- the ranks of a
//     job run one after the other on one GPU, each putting its chunk into its slot of the caller's next table;
//   * RCCL communicator (one process per GPU): a grouped ncclAllGather on the context's stream;
//   * group (one process, DMPC_DEVICE_ALL): every rank thread publishes where its next table is, meets the others at a host barrier,
//     copies its chunk into the slot of EVERY GPU's next table (hipMemcpyPeerAsync on its own stream: xGMI, the copy engines run
//     beside the kernels), records an event, meets the others again and makes its stream wait for their events -- whatever is
//     enqueued behind the exchange sees the whole table.  The table a peer writes into is never the one its owner's current step
//     reads (double buffering), and a rank only starts the step after next once it has waited for the events of this one: no
//     write ever overtakes a read.
static int exchange(dmpc_ctx *ctx, const void *chunk, void *table_next, size_t chunk_elems, int elem, const int *flags_loc, int *flags_all,
                    int flags_ints, hipStream_t st)
{
    const size_t cb = chunk_elems * (size_t)elem;
    if (ctx->grp) {
        GroupShared *sh = ctx->grp;
        const int G = sh->G, r = ctx->rank;
        sh->next_ptr[(size_t)r] = (double *)table_next; sh->fall_ptr[(size_t)r] = flags_all;
        if (!sh->wait()) FAIL(ctx, "group exchange: another rank failed");
        hipError_t e = hipSuccess;
        for (int q = 0; q < G && e == hipSuccess; ++q) {
            char *dst = (char *)sh->next_ptr[(size_t)q] + (size_t)r * cb;
            e = (sh->dev[(size_t)q] == ctx->device) ? hipMemcpyAsync(dst, chunk, cb, hipMemcpyDeviceToDevice, st)
                                                    : hipMemcpyPeerAsync(dst, sh->dev[(size_t)q], chunk, ctx->device, cb, st);
            if (e == hipSuccess && flags_loc) {
                int *fd = sh->fall_ptr[(size_t)q] + (size_t)r * flags_ints;
                e = (sh->dev[(size_t)q] == ctx->device) ? hipMemcpyAsync(fd, flags_loc, (size_t)flags_ints * 4, hipMemcpyDeviceToDevice, st)
                                                        : hipMemcpyPeerAsync(fd, sh->dev[(size_t)q], flags_loc, ctx->device, (size_t)flags_ints * 4, st);
            }
        }
        const int par = (int)(ctx->grp_steps++ & 1);
        if (e == hipSuccess) e = hipEventRecord(sh->ev[(size_t)r * 2 + par], st);
        if (e != hipSuccess) { sh->abort.store(1); FAIL(ctx, std::string("group exchange: ") + hipGetErrorString(e)); }
        if (!sh->wait()) FAIL(ctx, "group exchange: another rank failed");
        for (int q = 0; q < G; ++q)
            if (q != r) HIPCHK(ctx, hipStreamWaitEvent(st, sh->ev[(size_t)q * 2 + par], 0));
        return 0;
    }
    if (!ctx->comm) {
        HIPCHK(ctx, hipMemcpyAsync((char *)table_next + (size_t)ctx->rank * cb, chunk, cb, hipMemcpyDeviceToDevice, st));
        if (flags_loc) HIPCHK(ctx, hipMemcpyAsync(flags_all + (size_t)ctx->rank * flags_ints, flags_loc, (size_t)flags_ints * 4, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    mg::Api *a = mg::api();
    int rc = 0;
    if (flags_loc) rc = a->GroupStart();
    if (!rc) rc = (elem == 8) ? a->AllGather(chunk, table_next, chunk_elems, mg::kDouble, (mg::Comm)ctx->comm, st)
                              : a->AllGather(chunk, table_next, chunk_elems, mg::kFloat, (mg::Comm)ctx->comm, st);
    if (!rc && flags_loc) rc = a->AllGather(flags_loc, flags_all, (size_t)flags_ints, mg::kInt32, (mg::Comm)ctx->comm, st);
    if (flags_loc) { const int rc2 = a->GroupEnd(); if (!rc) rc = rc2; }
    if (rc) FAIL(ctx, std::string("ncclAllGather: ") + a->GetErrorString(rc));
    return 0;
}

extern "C" int dmpc_step_sharded_device(dmpc_ctx *ctx, int S, int N, const double *lT, const double *x_p, const double *x_v,
                                        const double *x_a, const double *pf, double *p_out, double *v_out, double *a_out,
                                        double *lT_next, int32_t *status, int32_t *info, void *stream)
{
    if (!ctx) { g_err = "dmpc_step_sharded_device: ctx is NULL"; return -1; }
    const int G = ctx->nranks, rank = ctx->rank;
    int32_t lo = 0, cnt = 0, cmax = 0;
    if (S < 1 || dmpc_partition(N, G, rank, &lo, &cnt, &cmax)) FAIL(ctx, "dmpc_step_sharded_device: bad S / N for this communicator");
    if (!lT || !x_p || !x_v || !x_a || !pf || !p_out || !v_out || !a_out || !lT_next || !status) FAIL(ctx, "dmpc_step_sharded_device: NULL pointer");
    if (ctx->grp) FAIL(ctx, "dmpc_step_sharded_device: device pointers belong to ONE GPU; a DMPC_DEVICE_ALL context drives several (use the host-pointer entry points)");
    if (G > 1 && !ctx->comm && !ctx->debug_rank) FAIL(ctx, "dmpc_step_sharded_device: no communicator");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t chunk = (size_t)S * N3 * cmax;
    if (ctx->sendbuf.ensure(chunk * 8)) FAIL(ctx, "device allocation failed (exchange buffer)");
    const int rem = N % G;
    // mixed precision: the caller's tables are fp64 (and so is the payload of this entry point); the scan reads an fp32 copy
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;
    if (mixed && table_f32(ctx, lT, ctx->lTf, (size_t)G * chunk, st)) return -1;
    if (launch_step(ctx, S, G, cmax, rank, 0, cnt, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, ctx->sendbuf.as<double>(), status, info, st,
                    nullptr, rem ? rem : 0, mixed ? ctx->lTf.as<float>() : nullptr))
        return -1;
    return exchange(ctx, ctx->sendbuf.p, lT_next, chunk, 8, nullptr, nullptr, 0, st);
}

// The scene-wide histories [S][N][KT][3] from the ranks' own ones (dmpc_transition_sharded leaves [S][cnt][KT][3] resident), into
// ctx->hist_* -- the input dmpc_postcheck takes when it is called without host arrays.  RCCL: every rank ends up with them
// (three all-gathers of padded slabs, once per transition); group: rank 0 does (each rank writes its slab into rank 0's arrays);
// single rank: nothing to do.
static int gather_histories(dmpc_ctx *ctx, int S, int N, int KT)
{
    const int G = ctx->nranks, rank = ctx->rank;
    if (G == 1) { ctx->hist_S = S; ctx->hist_N = N; ctx->hist_KT = KT; return 0; }
    int32_t lo = 0, cnt = 0, cmax = 0;
    if (dmpc_partition(N, G, rank, &lo, &cnt, &cmax)) FAIL(ctx, "gather_histories: bad partition");
    hipStream_t st = ctx->stream;
    const size_t per = (size_t)KT * 3, full = (size_t)S * N * per * 8;
    DevBuf *own[3] = {&ctx->hist_p, &ctx->hist_v, &ctx->hist_a}, *dstb[3] = {&ctx->full_p, &ctx->full_v, &ctx->full_a};
    if (ctx->grp) {
        GroupShared *sh = ctx->grp;
        dmpc_ctx *root = nullptr;
        // rank 0 allocates, publishes; everybody writes its slab (a kernel storing through the peer mapping, or a staged copy)
        if (rank == 0) {
            for (int u = 0; u < 3; ++u) if (dstb[u]->ensure(full)) { sh->abort.store(1); FAIL(ctx, "device allocation failed (gathered histories)"); }
            sh->hist_dst[0] = ctx->full_p.as<double>(); sh->hist_dst[1] = ctx->full_v.as<double>(); sh->hist_dst[2] = ctx->full_a.as<double>();
        }
        (void)root;
        if (!sh->wait()) FAIL(ctx, "group gather: another rank failed");
        const bool same = sh->dev[0] == ctx->device;
        int can = 1;
        if (!same && hipDeviceCanAccessPeer(&can, ctx->device, sh->dev[0]) != hipSuccess) can = 0;
        const size_t total = (size_t)S * cnt * per;
        const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        for (int u = 0; u < 3; ++u) {
            if (same || can)
                hipLaunchKernelGGL(mg::hist_place_kernel, dim3(blocks), dim3(256), 0, st, S, N, KT, (int)lo, (int)cnt, (int)cnt, (const double *)own[u]->p, sh->hist_dst[u]);
            else   // no peer mapping: one strided copy per array (rows = scenes)
                HIPCHK(ctx, hipMemcpy2DAsync(sh->hist_dst[u] + (size_t)lo * per, (size_t)N * per * 8, own[u]->p, (size_t)cnt * per * 8, (size_t)cnt * per * 8,
                                             (size_t)S, hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (!sh->wait()) FAIL(ctx, "group gather: another rank failed");
        if (rank == 0) {
            for (int u = 0; u < 3; ++u) std::swap(own[u]->p, dstb[u]->p), std::swap(own[u]->cap, dstb[u]->cap);
            ctx->hist_S = S; ctx->hist_N = N; ctx->hist_KT = KT;
        }
        return 0;
    }
    if (!ctx->comm) return 0;   // (emulated ranks without a communicator: nothing to gather from)
    mg::Api *a = mg::api();
    const size_t slab = (size_t)S * cmax * per;
    if (ctx->gath.ensure((size_t)G * slab * 8) || ctx->sendbuf.ensure(slab * 8)) FAIL(ctx, "device allocation failed (gathered histories)");
    for (int u = 0; u < 3; ++u) {
        if (dstb[u]->ensure(full)) FAIL(ctx, "device allocation failed (gathered histories)");
        const unsigned blocks = (unsigned)((slab + 255) / 256 > 8192 ? 8192 : (slab + 255) / 256);
        hipLaunchKernelGGL(mg::hist_pad_kernel, dim3(blocks), dim3(256), 0, st, S, KT, (int)cnt, (int)cmax, (const double *)own[u]->p, ctx->sendbuf.as<double>());
        const int rc = a->AllGather(ctx->sendbuf.p, ctx->gath.p, slab, mg::kDouble, (mg::Comm)ctx->comm, st);
        if (rc) FAIL(ctx, std::string("ncclAllGather (histories): ") + a->GetErrorString(rc));
        for (int g = 0; g < G; ++g) {
            int32_t glo = 0, gcnt = 0;
            (void)dmpc_partition(N, G, g, &glo, &gcnt, nullptr);
            const size_t tot = (size_t)S * gcnt * per;
            const unsigned bl = (unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256);
            hipLaunchKernelGGL(mg::hist_place_kernel, dim3(bl), dim3(256), 0, st, S, N, KT, (int)glo, (int)gcnt, (int)cmax,
                               (const double *)(ctx->gath.as<double>() + (size_t)g * slab), dstb[u]->as<double>());
        }
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int u = 0; u < 3; ++u) std::swap(own[u]->p, dstb[u]->p), std::swap(own[u]->cap, dstb[u]->cap);
    ctx->hist_S = S; ctx->hist_N = N; ctx->hist_KT = KT;
    return 0;
}

// the whole transition of S scenes for THIS rank's agents (see the header); gather != 0: the scene-wide histories are assembled
// afterwards (gather_histories: the input of dmpc_postcheck)
static int transition_sharded_impl(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                                   double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status, int gather)
{
    if (!ctx) { g_err = "dmpc_transition_sharded: ctx is NULL"; return -1; }
    const int G = ctx->nranks, rank = ctx->rank;
    int32_t lo = 0, cnt = 0, cmax = 0;
    if (S < 1 || K_T_max < 2 || !po || !pf || !K_T_used || !scene_status || ((pk || vk || ak) && !(pk && vk && ak)) ||
        dmpc_partition(N, G, rank, &lo, &cnt, &cmax))
        FAIL(ctx, "dmpc_transition_sharded: bad arguments");
    if (G > 1 && !ctx->comm && !ctx->grp && !ctx->debug_rank) FAIL(ctx, "dmpc_transition_sharded: no communicator");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;
    const size_t A = (size_t)S * N, Aown = (size_t)S * cnt, tab = (size_t)G * S * N3 * cmax, chunk = (size_t)S * N3 * cmax;
    const size_t hist = Aown * (size_t)K_T_max * 24;
    if (ensure_step_scratch(ctx, A, Aown)) return -1;
    if (ctx->lT.ensure(tab * 8) || ctx->lT2.ensure(tab * 8) || ctx->po.ensure(A * 24) || ctx->mg_pf.ensure(A * 24) || ctx->sendbuf.ensure(chunk * 8) ||
        ctx->hist_p.ensure(hist) || ctx->hist_v.ensure(hist) || ctx->hist_a.ensure(hist) || ctx->flags.ensure((size_t)K_T_max * S * 8) ||
        ctx->scene_done.ensure((size_t)S * 4) || ctx->mg_floc.ensure((size_t)S * 8) || ctx->mg_fall.ensure((size_t)G * S * 16) ||
        (mixed && (ctx->lTf.ensure(tab * 4) || ctx->lTf2.ensure(tab * 4) || ctx->sendbuf32.ensure(chunk * 4) || ctx->own64.ensure(chunk * 8))))
        FAIL(ctx, "device allocation failed");
    hipStream_t st = ctx->stream;
    double *xp = ctx->xp.as<double>(), *xv = ctx->xv.as<double>(), *xa = ctx->xa.as<double>(), *own_pf = ctx->pf.as<double>();
    HIPCHK(ctx, hipMemcpyAsync(ctx->po.p, po, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->mg_pf.p, pf, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->flags.p, 0, (size_t)K_T_max * S * 8, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->scene_done.p, 0, (size_t)S * 4, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_p.p, 0, hist, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_v.p, 0, hist, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_a.p, 0, hist, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->sendbuf.p, 0, chunk * 8, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->mg_fall.p, 0, (size_t)G * S * 16, st));
    // group transport: the peers PUSH their verdict flags into this rank's mg_fall right after the first barrier, on their own streams --
    // the memset above must have run by then
    if (ctx->grp) HIPCHK(ctx, hipStreamSynchronize(st));
    // k = 1: initDMPC for ALL agents (every rank builds the same first table), own states = (po, 0, 0)
    hipLaunchKernelGGL(init_rows_kernel, dim3((unsigned)((A * N3 + 255) / 256)), dim3(256), 0, st, (int)A, ctx->prm.h, ctx->po.as<double>(),
                       ctx->mg_pf.as<double>(), ctx->rows.as<double>());
    const int rem = N % G;
    {
        const unsigned blocks = (unsigned)((tab + 255) / 256 > 4096 ? 4096 : (tab + 255) / 256);
        hipLaunchKernelGGL(mg::table_from_rows_padded_kernel, dim3(blocks), dim3(256), 0, st, S, N, G, (int)cmax, rem, (const double *)ctx->rows.as<double>(),
                           ctx->lT.as<double>());
    }
    const unsigned sb = (unsigned)((Aown * 3 + 255) / 256);
    hipLaunchKernelGGL(mg::slice_agents_kernel, dim3(sb), dim3(256), 0, st, S, N, (int)lo, (int)cnt, (const double *)ctx->po.as<double>(), xp);
    hipLaunchKernelGGL(mg::slice_agents_kernel, dim3(sb), dim3(256), 0, st, S, N, (int)lo, (int)cnt, (const double *)ctx->mg_pf.as<double>(), own_pf);
    HIPCHK(ctx, hipMemsetAsync(xv, 0, Aown * 24, st));
    HIPCHK(ctx, hipMemsetAsync(xa, 0, Aown * 24, st));
    hipLaunchKernelGGL(record_kernel, dim3(sb), dim3(256), 0, st, S, (int)cnt, K_T_max, 0, xp, xv, xa, ctx->hist_p.as<double>(),
                       ctx->hist_v.as<double>(), ctx->hist_a.as<double>());
    // Mixed precision: the TABLE of the transition is fp32 -- the scan reads it, and what the ranks exchange per step is the fp32
    // chunk (half the payload); each rank keeps the fp64 predictions of its OWN chunk beside it (the solve's fallback for agents
    // that were not solved).  fp64: the table is the fp64 one and so is the payload.
    double *cur = ctx->lT.as<double>(), *nxt = ctx->lT2.as<double>();
    float *curf = ctx->lTf.as<float>(), *nxtf = ctx->lTf2.as<float>();
    double *own_cur = ctx->own64.as<double>();
    if (mixed) {
        if (table_f32(ctx, cur, ctx->lTf, tab, st)) return -1;
        curf = ctx->lTf.as<float>();
        HIPCHK(ctx, hipMemcpyAsync(own_cur, cur + (size_t)rank * chunk, chunk * 8, hipMemcpyDeviceToDevice, st));
    }
    // The gathered verdicts are double-buffered like the table: in the group transport a peer PUSHES its flags of step k+1 into this rank's
    // memory as soon as it has seen this rank's copies of step k -- possibly before this rank's combine kernel of step k has read the
    // buffer (found by running two groups side by side: a scene then stopped one step early on one rank).
    int *fall = ctx->mg_fall.as<int>();
    auto publish = [&](const int *floc) -> int {   // this step's predictions (ctx->sendbuf, fp64) into everybody's next table
        fall = ctx->mg_fall.as<int>() + (ctx->grp ? (size_t)(ctx->grp_steps & 1) * G * S * 2 : 0);
        if (!mixed) return exchange(ctx, ctx->sendbuf.p, nxt, chunk, 8, floc, fall, S * 2, st);
        const unsigned bl = (unsigned)((chunk + 255) / 256 > 4096 ? 4096 : (chunk + 255) / 256);
        hipLaunchKernelGGL(mg::chunk_to_f32_kernel, dim3(bl), dim3(256), 0, st, chunk, (const double *)ctx->sendbuf.as<double>(), ctx->sendbuf32.as<float>());
        return exchange(ctx, ctx->sendbuf32.p, nxtf, chunk, 4, floc, fall, S * 2, st);
    };
    // ReachedGoal on the initDMPC column (as dmpc_transition): own verdicts, exchanged with a throw-away table exchange
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->status.p, DMPC_ST_SOLVED, Aown, st));
    hipLaunchKernelGGL(scene_reduce_kernel, dim3((unsigned)S), dim3(256), 0, st, (int)cnt, error_tol, xp, own_pf, (const int *)ctx->status.as<int32_t>(),
                       ctx->mg_floc.as<int>(), (int *)nullptr);
    if (publish(ctx->mg_floc.as<int>())) return -1;
    hipLaunchKernelGGL(mg::combine_flags_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, G, S, (const int *)fall,
                       ctx->flags.as<int>(), ctx->scene_done.as<int>());
    std::vector<int32_t> flags((size_t)K_T_max * S * 2, 0);
    std::vector<int> done(S, 0);
    for (int s = 0; s < S; ++s) { K_T_used[s] = K_T_max; scene_status[s] = DMPC_ST_SOLVED; }
    int ndone = 0;
    const int chunk_steps = 8;   // the host looks at the per-step verdicts every 8 MPC steps
    for (int k = 1; k < K_T_max && ndone < S; ++k) {
        if (launch_step(ctx, S, G, cmax, rank, 0, cnt, cur, xp, xv, xa, own_pf, ctx->pout.as<double>(), ctx->vout.as<double>(),
                        ctx->aout.as<double>(), ctx->sendbuf.as<double>(), ctx->status.as<int32_t>(), nullptr, st, ctx->scene_done.as<int>(), rem,
                        mixed ? curf : nullptr, nullptr, mixed ? own_cur : nullptr)) {
            // a rank that cannot take its step must not leave the others waiting in the exchange: the group's barriers are released by
            // the abort flag, an RCCL communicator is aborted (the peers' collectives then return an error instead of hanging)
            if (ctx->grp) ctx->grp->abort.store(1);
            if (ctx->comm && mg::api()->CommAbort) { (void)mg::api()->CommAbort((mg::Comm)ctx->comm); ctx->comm = nullptr; ctx->nranks = 1; ctx->rank = 0; }
            return -1;
        }
        // state advance + history column + this rank's verdict per scene (all own agents at their goals / OR of their status
        // bits) in one launch, then the exchange: predictions into the next table, verdicts of all ranks next to them
        hipLaunchKernelGGL(post_step_kernel, dim3((unsigned)S), dim3(cnt >= 256 ? 256 : 128), 0, st, (int)cnt, K_T_max, k, error_tol,
                           (const double *)ctx->pout.as<double>(), (const double *)ctx->vout.as<double>(), (const double *)ctx->aout.as<double>(),
                           (const int *)ctx->status.as<int32_t>(), xp, xv, xa, (const double *)own_pf, ctx->hist_p.as<double>(), ctx->hist_v.as<double>(),
                           ctx->hist_a.as<double>(), ctx->mg_floc.as<int>(), (int *)nullptr, (const int *)ctx->scene_done.as<int>());
        HIPCHK(ctx, hipGetLastError());
        if (publish(ctx->mg_floc.as<int>())) return -1;
        if (mixed) HIPCHK(ctx, hipMemcpyAsync(own_cur, ctx->sendbuf.p, chunk * 8, hipMemcpyDeviceToDevice, st));   // own fp64 predictions of the new table
        hipLaunchKernelGGL(mg::combine_flags_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, G, S, (const int *)fall,
                           ctx->flags.as<int>() + (size_t)k * S * 2, ctx->scene_done.as<int>());
        std::swap(cur, nxt);   // l = new_l (dmpc_soft_bound.m:146)
        std::swap(curf, nxtf);
        if (k % chunk_steps == 0 || k == K_T_max - 1) {
            const int k0 = k <= chunk_steps ? 0 : ((k - 1) / chunk_steps) * chunk_steps + 1;
            HIPCHK(ctx, hipMemcpyAsync(&flags[(size_t)k0 * S * 2], ctx->flags.as<int>() + (size_t)k0 * S * 2, (size_t)(k - k0 + 1) * S * 8,
                                       hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            for (int kk = k0; kk <= k; ++kk)
                for (int s = 0; s < S; ++s) {
                    if (done[s]) continue;
                    const int32_t reached = flags[((size_t)kk * S + s) * 2], stbits = flags[((size_t)kk * S + s) * 2 + 1];
                    if (stbits & ~DMPC_ST_SOLVED) { done[s] = 1; ndone++; K_T_used[s] = kk + 1; scene_status[s] = stbits; }   // the same rule as dmpc_transition
                    else if (reached) { done[s] = 1; ndone++; K_T_used[s] = kk + 1; scene_status[s] = DMPC_ST_SOLVED | DMPC_ST_REACHED; }
                }
        }
    }
    if (pk) {
        HIPCHK(ctx, hipMemcpyAsync(pk, ctx->hist_p.p, hist, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(vk, ctx->hist_v.p, hist, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(ak, ctx->hist_a.p, hist, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->hist_S = 0;   // the resident histories hold only this rank's agents: not a dmpc_postcheck input ...
    if (gather) return gather_histories(ctx, S, N, K_T_max);   // ... unless they are assembled
    return 0;
}

extern "C" int dmpc_transition_sharded(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                                       double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status)
{
    if (ctx && ctx->grp) FAIL(ctx, "dmpc_transition_sharded: a DMPC_DEVICE_ALL context shards by itself (call dmpc_transition)");
    return transition_sharded_impl(ctx, S, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status, 0);
}

// same, and afterwards every rank assembles the scene-wide histories on its device (one all-gather per array): dmpc_postcheck with
// pk = NULL then checks the whole transition on any rank -- the post-checks of test/failure_rate.m:136-195 after a sharded run
extern "C" int dmpc_transition_sharded_gather(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                                              double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status)
{
    if (ctx && ctx->grp) FAIL(ctx, "dmpc_transition_sharded_gather: a DMPC_DEVICE_ALL context shards by itself (call dmpc_transition)");
    return transition_sharded_impl(ctx, S, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// one process, several GPUs (DMPC_DEVICE_ALL): the host-pointer entry points on a group context.  One host thread per rank for
// the duration of the call (the reference's DMPC::solveParallelDMPCv2 spawns and joins its cluster threads per MPC step,
// dmpc.cpp:1656-1686; here per call, the steps are joined on the devices).
// ---------------------------------------------------------------------------------------------------------------------------
template <class F>
static int group_run(dmpc_ctx *root, F &&body)
{
    GroupShared *sh = root->grp;
    const int G = sh->G;
    sh->abort.store(0); sh->arrived.store(0);
    // the parity of the event pairs and of the double-buffered verdict flags starts afresh: a call that failed half-way through an exchange
    // may have left the ranks' counters apart
    root->grp_steps = 0;
    for (dmpc_ctx *pc : root->peers) pc->grp_steps = 0;
    std::vector<int> rc((size_t)G, 0);
    std::vector<std::thread> th;
    for (int r = 1; r < G; ++r)
        th.emplace_back([&, r]() { rc[(size_t)r] = body(root->peers[(size_t)r - 1], r); if (rc[(size_t)r]) sh->abort.store(1); });
    rc[0] = body(root, 0);
    if (rc[0]) sh->abort.store(1);
    for (auto &t : th) t.join();
    (void)hipSetDevice(root->device);
    for (int r = 1; r < G; ++r)
        if (rc[(size_t)r] && !rc[0]) { root->err = "rank " + std::to_string(r) + ": " + root->peers[(size_t)r - 1]->err; g_err = root->err; return -1; }
    return rc[0] ? -1 : 0;
}

static int group_transition(dmpc_ctx *root, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol, double *pk,
                            double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status)
{
    const int G = root->grp->G;
    if (G > N) FAIL(root, "dmpc_transition: more GPUs than agents");
    std::vector<std::vector<int32_t>> used((size_t)G, std::vector<int32_t>((size_t)S)), sst((size_t)G, std::vector<int32_t>((size_t)S));
    const int rc = group_run(root, [&](dmpc_ctx *c, int r) -> int {
        return transition_sharded_impl(c, S, N, po, pf, K_T_max, error_tol, nullptr, nullptr, nullptr, r ? used[(size_t)r].data() : K_T_used,
                                       r ? sst[(size_t)r].data() : scene_status, 1);
    });
    if (rc) return -1;
    if (pk) {   // rank 0 holds the scene-wide histories now
        const size_t hist = (size_t)S * N * (size_t)K_T_max * 24;
        HIPCHK(root, hipSetDevice(root->device));
        HIPCHK(root, hipMemcpyAsync(pk, root->hist_p.p, hist, hipMemcpyDeviceToHost, root->stream));
        HIPCHK(root, hipMemcpyAsync(vk, root->hist_v.p, hist, hipMemcpyDeviceToHost, root->stream));
        HIPCHK(root, hipMemcpyAsync(ak, root->hist_a.p, hist, hipMemcpyDeviceToHost, root->stream));
        HIPCHK(root, hipStreamSynchronize(root->stream));
    }
    return 0;
}

// dmpc_step_batch on a group: every rank uploads the scene tables, solves its cluster, and writes its agents' rows of the outputs
static int group_step_batch(dmpc_ctx *root, int S, int N, const double *l, const double *x_p, const double *x_v, const double *x_a,
                            const double *pf, double *p_out, double *v_out, double *a_out, int32_t *status, int32_t *info)
{
    const int G = root->grp->G;
    if (G > N) FAIL(root, "dmpc_step_batch: more GPUs than agents");
    return group_run(root, [&](dmpc_ctx *c, int r) -> int {
        int32_t lo = 0, cnt = 0, cmax = 0;
        if (dmpc_partition(N, G, r, &lo, &cnt, &cmax)) return -1;
        HIPCHK(c, hipSetDevice(c->device));
        const size_t A = (size_t)S * N, Aown = (size_t)S * cnt, tab = (size_t)G * S * N3 * cmax;
        if (ensure_step_scratch(c, A, Aown) || c->lT.ensure(tab * 8) || c->po.ensure(A * 24)) FAIL(c, "device allocation failed");
        hipStream_t st = c->stream;
        HIPCHK(c, hipMemcpyAsync(c->rows.p, l, A * N3 * 8, hipMemcpyHostToDevice, st));
        const int rem = N % G;
        const unsigned blocks = (unsigned)((tab + 255) / 256 > 4096 ? 4096 : (tab + 255) / 256);
        hipLaunchKernelGGL(mg::table_from_rows_padded_kernel, dim3(blocks), dim3(256), 0, st, S, N, G, (int)cmax, rem, (const double *)c->rows.as<double>(),
                           c->lT.as<double>());
        // own agents' states: rows [lo, lo + cnt) of every scene (strided host arrays)
        const double *src[4] = {x_p, x_v, x_a, pf};
        double *dst[4] = {c->xp.as<double>(), c->xv.as<double>(), c->xa.as<double>(), c->pf.as<double>()};
        for (int u = 0; u < 4; ++u)
            HIPCHK(c, hipMemcpy2DAsync(dst[u], (size_t)cnt * 24, src[u] + (size_t)lo * 3, (size_t)N * 24, (size_t)cnt * 24, (size_t)S, hipMemcpyHostToDevice, st));
        const bool mixed = (c->precision & DMPC_PREC_MIXED) != 0;
        if (mixed && table_f32(c, c->lT.as<double>(), c->lTf, tab, st)) return -1;
        if (launch_step(c, S, G, cmax, r, 0, cnt, c->lT.as<double>(), dst[0], dst[1], dst[2], dst[3], c->pout.as<double>(), c->vout.as<double>(),
                        c->aout.as<double>(), nullptr, c->status.as<int32_t>(), c->info.as<int32_t>(), st, nullptr, rem, mixed ? c->lTf.as<float>() : nullptr))
            return -1;
        double *out[3] = {p_out, v_out, a_out};
        const double *dev[3] = {c->pout.as<double>(), c->vout.as<double>(), c->aout.as<double>()};
        for (int u = 0; u < 3; ++u)
            HIPCHK(c, hipMemcpy2DAsync(out[u] + (size_t)lo * N3, (size_t)N * N3 * 8, dev[u], (size_t)cnt * N3 * 8, (size_t)cnt * N3 * 8, (size_t)S, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpy2DAsync(status + lo, (size_t)N * 4, c->status.p, (size_t)cnt * 4, (size_t)cnt * 4, (size_t)S, hipMemcpyDeviceToHost, st));
        if (info) HIPCHK(c, hipMemcpy2DAsync(info + (size_t)lo * 8, (size_t)N * 32, c->info.p, (size_t)cnt * 32, (size_t)cnt * 32, (size_t)S, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        return 0;
    });
}
