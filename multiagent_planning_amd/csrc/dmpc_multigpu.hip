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

namespace mg {

typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;
enum { kDouble = 8, kInt32 = 2 };                  // ncclDataType_t: ncclInt32 = 2, ncclFloat64 = 8 (rccl.h)

struct Api {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};

static Api *api()
{
    static Api a;
    if (a.h || !a.err.empty()) return &a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) { a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.h) break; }
    if (!a.h) { a.err = std::string("cannot load librccl: ") + dlerror(); return &a; }
    auto sym = [&](const char *n) -> void * { void *p = dlsym(a.h, n); if (!p && a.err.empty()) a.err = std::string("librccl lacks ") + n; return p; };
    a.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
    a.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
    a.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    a.AllGather = (int (*)(const void *, void *, size_t, int, Comm, hipStream_t))sym("ncclAllGather");
    a.GroupStart = (int (*)())sym("ncclGroupStart");
    a.GroupEnd = (int (*)())sym("ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    return &a;
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
    if (!ctx || ctx->comm || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    ctx->nranks = nranks; ctx->rank = rank;
    return 0;
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
    ctx->nranks = 1; ctx->rank = 0;
    return 0;
}

// the exchange of one MPC step: every rank's chunk [S][3K][Cmax] into its slot of the next table (+ optionally the flags)
static int exchange(dmpc_ctx *ctx, const double *chunk, double *table_next, size_t chunk_doubles, const int *flags_loc, int *flags_all,
                    int flags_ints, hipStream_t st)
{
    if (!ctx->comm) {
        // no communicator: a single rank (the exchange is a device copy), or -- tests, dmpc_debug_set_rank -- the ranks of a
        // job run one after the other on one GPU, each putting its chunk into its slot of the caller's next table
        HIPCHK(ctx, hipMemcpyAsync(table_next + (size_t)ctx->rank * chunk_doubles, chunk, chunk_doubles * 8, hipMemcpyDeviceToDevice, st));
        if (flags_loc) HIPCHK(ctx, hipMemcpyAsync(flags_all + (size_t)ctx->rank * flags_ints, flags_loc, (size_t)flags_ints * 4, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    mg::Api *a = mg::api();
    int rc = 0;
    if (flags_loc) rc = a->GroupStart();
    if (!rc) rc = a->AllGather(chunk, table_next, chunk_doubles, mg::kDouble, (mg::Comm)ctx->comm, st);
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
    if (ctx->precision != DMPC_PREC_F64) FAIL(ctx, "dmpc_step_sharded_device: DMPC_PREC_F64 contexts only");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t chunk = (size_t)S * N3 * cmax;
    if (ctx->sendbuf.ensure(chunk * 8)) FAIL(ctx, "device allocation failed (exchange buffer)");
    const int rem = N % G;
    if (launch_step(ctx, S, G, cmax, rank, 0, cnt, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, ctx->sendbuf.as<double>(), status, info, st,
                    nullptr, rem ? rem : 0))
        return -1;
    return exchange(ctx, ctx->sendbuf.as<double>(), lT_next, chunk, nullptr, nullptr, 0, st);
}

// the whole transition of S scenes for THIS rank's agents (see the header)
extern "C" int dmpc_transition_sharded(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                                       double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status)
{
    if (!ctx) { g_err = "dmpc_transition_sharded: ctx is NULL"; return -1; }
    const int G = ctx->nranks, rank = ctx->rank;
    int32_t lo = 0, cnt = 0, cmax = 0;
    if (S < 1 || K_T_max < 2 || !po || !pf || !K_T_used || !scene_status || ((pk || vk || ak) && !(pk && vk && ak)) ||
        dmpc_partition(N, G, rank, &lo, &cnt, &cmax))
        FAIL(ctx, "dmpc_transition_sharded: bad arguments");
    if (ctx->precision != DMPC_PREC_F64) FAIL(ctx, "dmpc_transition_sharded: DMPC_PREC_F64 contexts only");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t A = (size_t)S * N, Aown = (size_t)S * cnt, tab = (size_t)G * S * N3 * cmax, chunk = (size_t)S * N3 * cmax;
    const size_t hist = Aown * (size_t)K_T_max * 24;
    if (ensure_step_scratch(ctx, A, Aown)) return -1;
    if (ctx->lT.ensure(tab * 8) || ctx->lT2.ensure(tab * 8) || ctx->po.ensure(A * 24) || ctx->mg_pf.ensure(A * 24) || ctx->sendbuf.ensure(chunk * 8) ||
        ctx->hist_p.ensure(hist) || ctx->hist_v.ensure(hist) || ctx->hist_a.ensure(hist) || ctx->flags.ensure((size_t)K_T_max * S * 8) ||
        ctx->scene_done.ensure((size_t)S * 4) || ctx->mg_floc.ensure((size_t)S * 8) || ctx->mg_fall.ensure((size_t)G * S * 8))
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
    double *cur = ctx->lT.as<double>(), *nxt = ctx->lT2.as<double>();
    // ReachedGoal on the initDMPC column (as dmpc_transition): own verdicts, exchanged with a throw-away table exchange
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->status.p, DMPC_ST_SOLVED, Aown, st));
    hipLaunchKernelGGL(scene_reduce_kernel, dim3((unsigned)S), dim3(256), 0, st, (int)cnt, error_tol, xp, own_pf, (const int *)ctx->status.as<int32_t>(),
                       ctx->mg_floc.as<int>(), (int *)nullptr);
    if (exchange(ctx, ctx->sendbuf.as<double>(), nxt, chunk, ctx->mg_floc.as<int>(), ctx->mg_fall.as<int>(), S * 2, st)) return -1;
    hipLaunchKernelGGL(mg::combine_flags_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, G, S, (const int *)ctx->mg_fall.as<int>(),
                       ctx->flags.as<int>(), ctx->scene_done.as<int>());
    std::vector<int32_t> flags((size_t)K_T_max * S * 2, 0);
    std::vector<int> done(S, 0);
    for (int s = 0; s < S; ++s) { K_T_used[s] = K_T_max; scene_status[s] = DMPC_ST_SOLVED; }
    int ndone = 0;
    const int chunk_steps = 8;   // the host looks at the per-step verdicts every 8 MPC steps
    for (int k = 1; k < K_T_max && ndone < S; ++k) {
        if (launch_step(ctx, S, G, cmax, rank, 0, cnt, cur, xp, xv, xa, own_pf, ctx->pout.as<double>(), ctx->vout.as<double>(),
                        ctx->aout.as<double>(), ctx->sendbuf.as<double>(), ctx->status.as<int32_t>(), nullptr, st, ctx->scene_done.as<int>(), rem))
            return -1;
        // state advance + history column + this rank's verdict per scene (all own agents at their goals / OR of their status
        // bits) in one launch, then the exchange: predictions into the next table, verdicts of all ranks next to them
        hipLaunchKernelGGL(post_step_kernel, dim3((unsigned)S), dim3(cnt >= 256 ? 256 : 128), 0, st, (int)cnt, K_T_max, k, error_tol,
                           (const double *)ctx->pout.as<double>(), (const double *)ctx->vout.as<double>(), (const double *)ctx->aout.as<double>(),
                           (const int *)ctx->status.as<int32_t>(), xp, xv, xa, (const double *)own_pf, ctx->hist_p.as<double>(), ctx->hist_v.as<double>(),
                           ctx->hist_a.as<double>(), ctx->mg_floc.as<int>(), (int *)nullptr, (const int *)ctx->scene_done.as<int>());
        HIPCHK(ctx, hipGetLastError());
        if (exchange(ctx, ctx->sendbuf.as<double>(), nxt, chunk, ctx->mg_floc.as<int>(), ctx->mg_fall.as<int>(), S * 2, st)) return -1;
        hipLaunchKernelGGL(mg::combine_flags_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, G, S, (const int *)ctx->mg_fall.as<int>(),
                           ctx->flags.as<int>() + (size_t)k * S * 2, ctx->scene_done.as<int>());
        std::swap(cur, nxt);   // l = new_l (dmpc_soft_bound.m:146)
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
    ctx->hist_S = 0;   // the resident histories hold only this rank's agents: not a dmpc_postcheck input
    return 0;
}
