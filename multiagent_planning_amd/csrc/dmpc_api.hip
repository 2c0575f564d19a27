// dmpc_api.hip -- host side of libdmpc_hip.so: the C ABI declared in include/dmpc_hip.h.
//
// Owns the HIP context state (stream, precomputed per-case tables, scratch buffers) and launches
// the kernels of dmpc_kernels.hip.  There is deliberately NO CPU fallback anywhere in this file:
// without a HIP device every compute entry point fails with an error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dmpc_hip.h"
#include "../../include/dmpc_hip_dev.h"
#include "dmpc_device.h"

#include "dmpc_kernels.hip"
#include "dmpc_postcheck.hip"
#include "dmpc_rowbuild.hip"
#include "dmpc_generators.hip"   // single translation unit: kernels + host ABI

using namespace dmpc;

static thread_local std::string g_err;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) return -1;
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    ~DevBuf() { release(); }   // every buffer of a context goes with it (dmpc_destroy selects the device first)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <class T> T *as() { return (T *)p; }
};

// One process, several GPUs (dmpc_create(prm, DMPC_DEVICE_ALL, ..)): what the sub-contexts of a group share -- a host barrier for
// their threads, the pointers each rank publishes for the per-step exchange, and the events that order the peer copies.
struct GroupShared {
    int G = 1;
    std::atomic<int> arrived{0};
    std::atomic<int> phase{0};
    std::atomic<int> abort{0};          // a rank failed: everybody leaves the barriers with an error instead of waiting forever
    std::vector<double *> next_ptr;     // [G] table the rank reads in the NEXT step (peers write their chunks into it)
    std::vector<int *> fall_ptr;        // [G] the rank's gathered scene verdicts
    std::vector<int> dev;               // [G] HIP device of the rank
    std::vector<hipEvent_t> ev;         // [G][2] "rank's copies of this step are enqueued", alternating
    double *hist_dst[3] = {nullptr, nullptr, nullptr};   // rank 0's scene-wide history arrays (gather_histories)
    // sense-reversing spin barrier of the G rank threads; false: the group was aborted
    bool wait()
    {
        const int ph = phase.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == G - 1) {
            arrived.store(0, std::memory_order_relaxed);
            phase.store(ph + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (phase.load(std::memory_order_acquire) == ph) {
                if (abort.load(std::memory_order_relaxed)) return false;
                if (++spins > 256) std::this_thread::yield();
            }
        }
        return !abort.load(std::memory_order_relaxed);
    }
};

struct dmpc_ctx {
    int device = 0;
    int precision = DMPC_PREC_F64;   // DMPC_PREC_MIXED: fp32 table / scan / rows, fp64 QP (host-pointer entry points)
    hipStream_t stream = nullptr;
    dmpc_params prm{};
    double *d_tables = nullptr;   // [3][900] Gram tables + [225] Lambda' table (dmpc_device.h: TAB_DOUBLES)
    double hsum[3] = {0, 0, 0};   // per cost case: sum of |H1(i,j)| (dual-bound certificate of the slack-free variants)
    std::string err;
    int64_t solves = 0;
    int max_lds_scp = 0;
    int max_lds_set = 0;
    // scratch for the host-pointer entry points
    DevBuf post_acc; int post_acc_S = 0, post_fused = 0;
    int no_fuse = 0;         // development option no_fuse: always launch post_step_kernel
    DevBuf rowbuf, rowkc, hdr, order, bbox, bbox_nm, nbr_list, nbr_cnt, lrow, counter, flag_list, scene_done;
    int num_cu = 0;
    int no_persist = 0;      // development option no_persist: one-agent-per-workgroup solve launches
    int max_lds_persist = 0;
    int force_persist = 0;   // development option force_persist (tests): the persistent kernel on small launches
    int cull_min = 256;      // development option cull_min: neighbour lists from this many agents per scene on
    int order_slices = 0;    // development option order_slices: workgroups of the order kernel (0: by launch size)
    int no_cull = 0;         // development option no_cull: no neighbour lists in the scan of large scenes (A/B runs, tests)
    int no_lpt = 0;          // development option no_lpt: no heaviest-first solve order
    int order_hint = 0;      // option order_hint = 1: the launch order also uses the agents' work estimates of the context's previous step (measured: no gain in
                             // closed loops -- the heavy agents of a step are not the heavy agents of the step before -- so off; a replay of ONE step would flatter it)
    DevBuf prev_cost;        // [S * c_count] work estimates of the previous step (solve kernel -> order kernel)
    long prev_cost_shape = -1;
    int crash_min = CRASH_MIN_DEFAULT;   // see StepParams::crash_min (development option crash_min; crash_any: also for the slack-free variants)
    int crash_any = 0;
    int pivot_explore = 0;   // development option pivot_explore (DMPC_PIVOT_EXPLORE builds)
    int no_fast_exit = 0;    // development option no_fast_exit (tests): every agent through the solve kernel (the unconstrained exit of the scan off)
    int iter_cap = ITER_CAP; // development option iter_cap: cap of the active-set iterations (agents beyond it end DMPC_ST_ITERCAP)
    int tier1_env = 0;       // development option tier1_qcap (tests): 32 = two tiers for the slack variants (any value: no shallow-launch shortcut)
    int single_tier = 0;         // 1: solve with the full working-set capacity in one launch
    DevBuf lTf, lTf2;            // mixed precision: fp32 copies of the tables the scan reads
    DevBuf rows, lT, lT2, xp, xv, xa, pf, po, pout, vout, aout, status, info, hist_p, hist_v, hist_a, flags;
    int hist_S = 0, hist_N = 0, hist_KT = 0;   // shape of the histories left resident by the last dmpc_transition
    int32_t *flags_host = nullptr; size_t flags_host_cap = 0;   // pinned: per-step verdicts of dmpc_transition
    hipEvent_t flag_ev[2] = {nullptr, nullptr};
    int pc_fallback_scenes = 0;   // last dmpc_postcheck: scenes of a cell-grid search that were searched again by brute force
    std::vector<dmpc_ctx *> children;   // further contexts (own stream and buffers) for the other parts of a split batch of transitions
    std::vector<int> split_at;          // non-empty: the last dmpc_transition left scenes [split_at[i], split_at[i+1]) in part i (0: here, i > 0: children[i-1])
    int split_parts = 0;     // development option split_parts: number of parts (0: the built-in rule)
    int no_split = 0;        // development option no_split
    std::string last_kernel; // the solve kernel the last step launched for the bulk of its agents (dmpc_last_solve_kernel)
    int reduced_solver = 1;  // solveSoftDMPCbound: the reduced solver (dmpc_rsolve.hip) in front of the general one; 0: the general solver alone (A/B runs, tests)
    int rsolve_cap = 0;      // development option rsolve_cap: the reduced solver hands an agent over after this many equality solves of a ladder level (0: its default; tests of the hand-over)
    int rsolve_blocks = 0;   // workgroups of dmpc_rsolve_persist_kernel a CU holds (occupancy query, once per context)
    int no_split_t = 0;      // development option no_split_t: slack-free persistent solve with the whole inverse factor in every wave's block (nine waves per CU; A/B runs, tests)
    int grid_min = 768;      // development option grid_min: cell-grid neighbour lists from this many agents per scene on (below: nbr_kernel) ...
    int grid_min_part = 2048; // ... and when the query covers only a PART of the scene's agents (a rank's chunk: the grid is still built over all of them)
    int prep_fuse = 1;       // development option prep_fuse: 0 = the cell grid of a single scene by the five kernels of round 4 instead of grid_prep_kernel + grid_fill2_kernel
    bool grid_clean = false; // the cell grid's counters were left zero by the last scan launch (grid_clean_key: for which buffer / size)
    unsigned long long grid_clean_key = 0;
    int nbr_grid = 1;        // development option nbr_grid: 0 = neighbour lists of large scenes from the all-pairs box test of round 3 (nbr_kernel) instead of the cell grid + distance filter
    DevBuf grid;             // cell grid of the neighbour lists (counts, starts, entries)
    int no_level_skip = 0;   // development option no_level_skip (see StepParams)
    int no_level_check = 0;  // development option no_level_check (see StepParams)
    int lds_pad_kb = 0;      // development option lds_pad_kb: KB of unused LDS per one-agent solve workgroup (occupancy experiments: fewer resident agents per CU)
    int f32_dep_exp = 8;     // development option f32_dep_exp: fp32-factor kernels treat a pivot as dependent below delta / s_pp = 10^-n
    int ext_cap = 0;         // development option ext_cap (tests): at most this many T extensions per workgroup (1: every agent that needs one waits for the same slot)
    int queue_chunk = 0;     // development option queue_chunk: positions per ticket of the persistent queue's light bulk (0: chosen per launch)
    int static_queue = 0;    // development option static_queue: persistent waves take queue positions round-robin instead of by ticket
    DevBuf pc_p, pc_v, pc_a, pc_M, pc_w, pc_scene, pc_agent, pc_interp;   // post-check work buffers
    DevBuf pc_pts, pc_cell, pc_fill, pc_start, pc_sorted, pc_on;           // post-check, large scenes: cell grid of a batch of samples
    // multi-GPU (dmpc_multigpu.hip): RCCL communicator of this rank, exchange buffers
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    DevBuf sendbuf, sendbuf32, own64, mg_pf, mg_floc, mg_fall, full_p, full_v, full_a, gath;
    // one process, several GPUs: this context is rank `rank` of the group `grp`; the context the caller holds (rank 0) owns
    // the others (`peers`, ranks 1..G-1) and the shared block
    GroupShared *grp = nullptr;
    std::vector<dmpc_ctx *> peers;
    long grp_steps = 0;      // exchanges of this rank so far (parity of the event pair)
    int grp_emulated = 0;    // the group's ranks share one device (dmpc_debug_emulate_devices): a second group is built the same way
    int debug_rank = 0;      // dmpc_debug_set_rank: emulated rank without a transport (tests)
    DevBuf rb_A, rb_l, rb_sel, rb_out, rb_bin, rb_po, gen_out, hp_in, hp_out;                     // dense row builders (host-pointer entries)
    // profiling
    int profile = 0;
    double *dbg = nullptr; int dbg_agent = -1, dbg_cap = 0;   // development trace (dmpc_debug_trace)
    DevBuf forced_order; int forced_n = 0;                    // development aid (dmpc_debug_set_order)
    struct Ev { hipEvent_t t0, t1, t2; };   // step start | scan+order done | solve tiers done
    std::vector<Ev> events, ev_pool;
    double prof_scan_ms_sum = 0.0;
    double prof_ms_sum = 0.0;
    int64_t prof_n = 0;
};

#define FAIL(ctx, msg)                                   \
    do {                                                 \
        std::string m_ = (msg);                          \
        if (ctx) (ctx)->err = m_;                        \
        g_err = m_;                                      \
        return -1;                                       \
    } while (0)
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) FAIL(ctx, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

// ---------------------------------------------------------------------------------------------
// host math: model matrices (a1-a3) and the per-case tables
// ---------------------------------------------------------------------------------------------

static int check_params(const dmpc_params *p, std::string &why)
{
    if (!p) { why = "params is NULL"; return -1; }
    if (p->K != K) { why = "only K = k_hor = 15 is supported (the value every reference script uses)"; return -1; }
    if (p->order != 2 && !(p->order == 4 && (p->variant == DMPC_VAR_SOFTALL || p->variant == DMPC_VAR_SOFTALL_C || p->variant == DMPC_VAR_ELLIP || p->variant == DMPC_VAR_REPAIR || p->variant == DMPC_VAR_CPP1))) {
        why = "ellipsoid order must be 2, or 4 with solveSoftDMPC / solveEllipDMPC / solveSoftDMPCrepair / DMPC::solveQP (test/comp_test_ellipconstr.m:158)"; return -1;
    }
    if (p->variant < 0 || p->variant > DMPC_VAR_SCP) { why = "unknown variant"; return -1; }
    if (p->variant == DMPC_VAR_SCP && !(p->tol >= 0)) { why = "solveDMPC: tol must be >= 0"; return -1; }
    if (!(p->h > 0) || !(p->rmin > 0) || !(p->c > 0) || !(p->alim > 0)) { why = "h, rmin, c, alim must be positive"; return -1; }
    for (int d = 0; d < 3; ++d)
        if (!(p->pmax[d] > p->pmin[d])) { why = "pmax must exceed pmin"; return -1; }
    return 0;
}

extern "C" int dmpc_model_matrices(const dmpc_params *prm, double *Lambda, double *Av, double *A0, double *Delta)
{
    if (!prm || prm->K < 1 || prm->K > 64) { g_err = "dmpc_model_matrices: bad params"; return -1; }
    const int Kh = prm->K, n = 3 * Kh;
    const double h = prm->h;
    // Same floating-point recurrences as the reference so the matrices agree bit for bit:
    //   row_k = Aux*row_{k-1} + [0 .. b .. 0]  (getPosMat.m:18-23, dmpc_soft_bound.m:100-105)
    //   A_init = Aux*A_init                     (dmpc_soft_bound.m:106-107)
    // with Aux = [I3 h*I3; 0 I3], b = [h^2/2*I3; h*I3].  Per axis the state is (pos,vel) so only the
    // two scalar sequences pr[j], vr[j] (coefficients of a_j) are needed.
    std::vector<double> pr(Kh, 0.0), vr(Kh, 0.0);
    double a0p = 1.0, a0v = 0.0;   // A_init(1,1), A_init(1,4) of the current power of Aux
    if (Lambda) memset(Lambda, 0, sizeof(double) * n * n);
    if (Av) memset(Av, 0, sizeof(double) * n * n);
    if (A0) memset(A0, 0, sizeof(double) * n * 6);
    for (int k = 0; k < Kh; ++k) {
        for (int j = 0; j < Kh; ++j) {
            const double np_ = 1.0 * pr[j] + h * vr[j];   // Aux rows 1:3
            const double nv_ = 1.0 * vr[j];               // Aux rows 4:6
            pr[j] = np_; vr[j] = nv_;
        }
        pr[k] += h * h / 2; vr[k] += h;
        a0v = 1.0 * a0v + h * 1.0;   // (Aux*A_init)(1,4) = A_init(1,4) + h*A_init(4,4)
        for (int j = 0; j < Kh; ++j)
            for (int a = 0; a < 3; ++a) {
                if (Lambda) Lambda[(size_t)(3 * k + a) * n + 3 * j + a] = pr[j];
                if (Av) Av[(size_t)(3 * k + a) * n + 3 * j + a] = vr[j];
            }
        if (A0)
            for (int a = 0; a < 3; ++a) { A0[(size_t)(3 * k + a) * 6 + a] = a0p; A0[(size_t)(3 * k + a) * 6 + 3 + a] = a0v; }
    }
    if (Delta) {   // getDeltaMat.m:2-8: [I 0 ..; -I I 0 ..; ...]
        memset(Delta, 0, sizeof(double) * n * n);
        for (int k = 0; k < Kh; ++k)
            for (int a = 0; a < 3; ++a) {
                Delta[(size_t)(3 * k + a) * n + 3 * k + a] = 1.0;
                if (k > 0) Delta[(size_t)(3 * k + a) * n + 3 * (k - 1) + a] = -1.0;
            }
    }
    return 0;
}

extern "C" int dmpc_posvel_matrix(double h, int Kh, double *Aaug)
{
    // getPosVelMat.m:24: Aaug = [new_row(K); [0 .. I3]; [I3 0 ..]]  (12 x 3K): final position and
    // velocity rows, then the selectors of the last and first acceleration
    if (!Aaug || Kh < 1) { g_err = "dmpc_posvel_matrix: bad arguments"; return -1; }
    const int n = 3 * Kh;
    memset(Aaug, 0, sizeof(double) * 12 * n);
    for (int j = 0; j < Kh; ++j)
        for (int a = 0; a < 3; ++a) {
            Aaug[(size_t)a * n + 3 * j + a] = h * h / 2 + (double)(Kh - 1 - j) * h * h;
            Aaug[(size_t)(3 + a) * n + 3 * j + a] = h;
        }
    for (int a = 0; a < 3; ++a) {
        Aaug[(size_t)(6 + a) * n + 3 * (Kh - 1) + a] = 1.0;
        Aaug[(size_t)(9 + a) * n + a] = 1.0;
    }
    return 0;
}

// H1 = 2(q l_K l_K' + s D1'D1 + I)  (per-axis block of solveSoftDMPCbound.m:98); returns
// H1^-1, M1 = H1^-1 L', P1 = L H1^-1 L' (row-major 15x15 each)
static void build_case_tables(double h, double q, double s, double *out /*675*/, double *hsum = nullptr)
{
    long double L[K][K], H[K][K], Hi[K][K], C[K][K];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) L[i][j] = (j <= i) ? ((long double)h * h / 2 + (long double)(i - j) * h * h) : 0.0L;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            long double dd = 0.0L;   // (D1'D1)_{ij}: tridiagonal [.. -1 2 -1 ..], last diagonal entry 1
            if (i == j) dd = (i == K - 1) ? 1.0L : 2.0L;
            else if (i == j + 1 || j == i + 1) dd = -1.0L;
            H[i][j] = 2.0L * ((long double)q * L[K - 1][i] * L[K - 1][j] + (long double)s * dd + (i == j ? 1.0L : 0.0L));
        }
    if (hsum) {   // sum of |H(i,j)|, rounded up: max over |a| <= alim of a'Ha/2 is at most alim^2/2 times this
        long double t = 0.0L;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) t += fabsl(H[i][j]);
        *hsum = (double)(t * (1.0L + 1e-12L));
    }
    // Cholesky H = C C'
    memset(C, 0, sizeof(C));
    for (int j = 0; j < K; ++j) {
        long double d = H[j][j];
        for (int k = 0; k < j; ++k) d -= C[j][k] * C[j][k];
        C[j][j] = sqrtl(d);
        for (int i = j + 1; i < K; ++i) {
            long double t = H[i][j];
            for (int k = 0; k < j; ++k) t -= C[i][k] * C[j][k];
            C[i][j] = t / C[j][j];
        }
    }
    // Hi = H^-1 column by column
    for (int c = 0; c < K; ++c) {
        long double y[K], x[K];
        for (int i = 0; i < K; ++i) {
            long double t = (i == c) ? 1.0L : 0.0L;
            for (int k = 0; k < i; ++k) t -= C[i][k] * y[k];
            y[i] = t / C[i][i];
        }
        for (int i = K - 1; i >= 0; --i) {
            long double t = y[i];
            for (int k = i + 1; k < K; ++k) t -= C[k][i] * x[k];
            x[i] = t / C[i][i];
        }
        for (int i = 0; i < K; ++i) Hi[i][c] = x[i];
    }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            long double sym = 0.5L * (Hi[i][j] + Hi[j][i]);
            out[i * K + j] = (double)sym;
        }
    long double M[K][K];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            long double t = 0.0L;
            for (int k = 0; k < K; ++k) t += 0.5L * (Hi[i][k] + Hi[k][i]) * L[j][k];
            M[i][j] = t;
            out[225 + i * K + j] = (double)t;
        }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            long double t = 0.0L;
            for (int k = 0; k < K; ++k) t += L[i][k] * M[k][j];
            out[450 + i * K + j] = (double)t;
        }
    // P1 is symmetric in exact arithmetic: symmetrise the rounded table
    for (int i = 0; i < K; ++i)
        for (int j = i + 1; j < K; ++j) {
            double m = 0.5 * (out[450 + i * K + j] + out[450 + j * K + i]);
            out[450 + i * K + j] = out[450 + j * K + i] = m;
        }
}

// the device tables: per cost case the symmetric 30x30 Gram table over (space, step) -- G[A i][A j] = H1^-1, G[A i][W j] =
// (H1^-1 L')(i,j), G[W i][W j] = L H1^-1 L' -- then Lt[k][kk] = Lambda(kk,k)
static int upload_tables(dmpc_ctx *ctx)
{
    std::vector<double> t(TAB_ALL_DOUBLES, 0.0);
    const dmpc_params &p = ctx->prm;
    const double sfree = p.Sfree > 0 ? p.Sfree : 10.0;
    const double qs[3] = {p.Qfar > 0 ? p.Qfar : 1000.0, p.Qnear > 0 ? p.Qnear : 10000.0, p.Q1};   // far (:44-47), near (:49-52), coll (:54-57)
    const double ss[3] = {sfree, sfree, (p.variant == DMPC_VAR_ALL3) ? 10.0 : p.S1};             // all:71
    for (int c = 0; c < 3; ++c) {
        double hmp[675];
        build_case_tables(p.h, qs[c], ss[c], hmp, &ctx->hsum[c]);
        double *G = &t[(size_t)c * TAB_CASE_DOUBLES];
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                G[i * 30 + j] = hmp[i * K + j];
                G[i * 30 + 15 + j] = hmp[225 + i * K + j];
                G[(15 + j) * 30 + i] = hmp[225 + i * K + j];
                G[(15 + i) * 30 + 15 + j] = hmp[450 + i * K + j];
            }
        // Tp = C^-T with H1^-1 = C C' (the ROUNDED table above: the numbers the solver's Schur complement is made of).  The leading
        // m x m block of Tp is the inverse factor of the leading block of H1^-1: S^-1 = Tp Tp' for the bounds of steps 0..m-1 of one axis.
        // Second table: the same for the steps in FALLING order (14, 13, ..): H1^-1 with rows and columns reversed.
        for (int rev = 0; rev < 2; ++rev) {
            long double C[K][K], Ci[K][K];
            memset(C, 0, sizeof(C)); memset(Ci, 0, sizeof(Ci));
            auto hi = [&](int i, int j) -> long double { return rev ? hmp[(K - 1 - i) * K + (K - 1 - j)] : hmp[i * K + j]; };
            for (int j = 0; j < K; ++j) {
                long double d = hi(j, j);
                for (int k = 0; k < j; ++k) d -= C[j][k] * C[j][k];
                C[j][j] = sqrtl(d);
                for (int i = j + 1; i < K; ++i) {
                    long double v = hi(i, j);
                    for (int k = 0; k < j; ++k) v -= C[i][k] * C[j][k];
                    C[i][j] = v / C[j][j];
                }
            }
            for (int c2 = 0; c2 < K; ++c2)   // C^-1 column by column (forward substitution)
                for (int i = c2; i < K; ++i) {
                    long double v = (i == c2) ? 1.0L : 0.0L;
                    for (int k = c2; k < i; ++k) v -= C[i][k] * Ci[k][c2];
                    Ci[i][c2] = v / C[i][i];
                }
            double *Tp = &t[(size_t)TAB_DOUBLES + (size_t)(2 * c + rev) * TAB_TP_CASE];
            for (int i = 0; i < K; ++i)
                for (int j = i; j < K; ++j) Tp[i * (31 - i) / 2 + j - i] = (double)Ci[j][i];
        }
    }
    for (int k = 0; k < K; ++k)
        for (int kk = k; kk < K; ++kk) t[3 * TAB_CASE_DOUBLES + k * K + kk] = p.h * p.h / 2 + (double)(kk - k) * p.h * p.h;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_tables, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------

extern "C" const char *dmpc_last_error(const dmpc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }
extern "C" int dmpc_abi_version(void) { return DMPC_ABI_VERSION; }
extern "C" const char *dmpc_last_solve_kernel(const dmpc_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

static std::atomic<int> g_emulate_devices{0};
// development / tests (not in the public header): DMPC_DEVICE_ALL then builds a group of n ranks that all sit on the calling thread's
// current device -- the single-process multi-GPU path (threads, peer copies, events) on a box with ONE GPU.  n = 0: off.
extern "C" int dmpc_debug_emulate_devices(int n)
{
    if (n < 0 || n > 64) return -1;
    g_emulate_devices.store(n);
    return 0;
}

// Development / test options of a context (not in the public header; nothing in the library reads the environment per call).  They select
// launch forms and tiers, never arithmetic: every combination returns the same bits (tests/test_gpu_paths.py).  A process can preset them for
// the contexts it creates with ONE environment variable, DMPC_DEBUG_OPTIONS="name=value,name=value" (the probes under tools/).
extern "C" int dmpc_debug_option(dmpc_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return -1;
    struct { const char *n; int dmpc_ctx::*f; } tab[] = {
        {"no_fuse", &dmpc_ctx::no_fuse}, {"no_persist", &dmpc_ctx::no_persist}, {"force_persist", &dmpc_ctx::force_persist}, {"no_cull", &dmpc_ctx::no_cull}, {"order_slices", &dmpc_ctx::order_slices}, {"cull_min", &dmpc_ctx::cull_min},
        {"no_lpt", &dmpc_ctx::no_lpt}, {"order_hint", &dmpc_ctx::order_hint}, {"crash_min", &dmpc_ctx::crash_min}, {"crash_any", &dmpc_ctx::crash_any}, {"no_fast_exit", &dmpc_ctx::no_fast_exit}, {"pivot_explore", &dmpc_ctx::pivot_explore},
        {"iter_cap", &dmpc_ctx::iter_cap}, {"tier1_qcap", &dmpc_ctx::tier1_env}, {"split_parts", &dmpc_ctx::split_parts}, {"no_split", &dmpc_ctx::no_split},
        {"no_level_skip", &dmpc_ctx::no_level_skip}, {"prep_fuse", &dmpc_ctx::prep_fuse}, {"static_queue", &dmpc_ctx::static_queue}, {"queue_chunk", &dmpc_ctx::queue_chunk}, {"no_split_t", &dmpc_ctx::no_split_t}, {"ext_cap", &dmpc_ctx::ext_cap}, {"nbr_grid", &dmpc_ctx::nbr_grid}, {"f32_dep_exp", &dmpc_ctx::f32_dep_exp}, {"grid_min", &dmpc_ctx::grid_min}, {"no_level_check", &dmpc_ctx::no_level_check}, {"lds_pad_kb", &dmpc_ctx::lds_pad_kb}, {"reduced_solver", &dmpc_ctx::reduced_solver}, {"rsolve_cap", &dmpc_ctx::rsolve_cap}};
    for (auto &t : tab)
        if (!std::strcmp(t.n, name)) {
            ctx->*(t.f) = value;
            if (t.f == &dmpc_ctx::grid_min) ctx->grid_min_part = value;   // (the option forces the grid for every query from that size on)
            for (dmpc_ctx *pc : ctx->peers) (void)dmpc_debug_option(pc, name, value);
            for (dmpc_ctx *ch : ctx->children) (void)dmpc_debug_option(ch, name, value);
            return 0;
        }
    ctx->err = std::string("dmpc_debug_option: unknown option ") + name;
    return -1;
}
static void options_from_env(dmpc_ctx *ctx)
{
    const char *e = getenv("DMPC_DEBUG_OPTIONS");
    if (!e) return;
    std::string all(e);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string item = all.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) (void)dmpc_debug_option(ctx, item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1));
        pos = end + 1;
    }
}

static dmpc_ctx *create_one(const dmpc_params *prm, int device, int precision)
{
    if (hipSetDevice(device) != hipSuccess) { g_err = "dmpc_create: hipSetDevice failed"; return nullptr; }
    dmpc_ctx *ctx = new dmpc_ctx();
    ctx->device = device;
    ctx->precision = precision;
    ctx->prm = *prm;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&ctx->d_tables, sizeof(double) * TAB_ALL_DOUBLES) != hipSuccess || upload_tables(ctx) != 0) {
        g_err = "dmpc_create: device initialisation failed: " + ctx->err;
        dmpc_destroy(ctx);
        return nullptr;
    }
    if (hipDeviceGetAttribute(&ctx->num_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->num_cu = 0;
    options_from_env(ctx);
    return ctx;
}

extern "C" dmpc_ctx *dmpc_create(const dmpc_params *prm, int device, int precision)
{
    std::string why;
    if (check_params(prm, why)) { g_err = "dmpc_create: " + why; return nullptr; }
    if (precision < DMPC_PREC_F64 || precision > DMPC_PREC_LOW) { g_err = "dmpc_create: precision must be DMPC_PREC_F64, _MIXED, _F32FACTOR or _LOW"; return nullptr; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = std::string("dmpc_create: no HIP device available (") + hipGetErrorString(e) +
                "); this library has no CPU fallback";
        return nullptr;
    }
    if (device == DMPC_DEVICE_CURRENT && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device < 0 && device != DMPC_DEVICE_ALL) { g_err = "dmpc_create: device must be a HIP device index, DMPC_DEVICE_CURRENT or DMPC_DEVICE_ALL"; return nullptr; }
    if (device == DMPC_DEVICE_ALL) {
        // every visible GPU, one process: rank r of the group on device r (the reference's thread clusters, dmpc.cpp:1600-1625,
        // one GPU each).  One visible GPU: a plain context.
        std::vector<int> devs;
        const int emu = g_emulate_devices.load();
        if (emu > 0) {
            int cur = 0;
            if (hipGetDevice(&cur) != hipSuccess) cur = 0;
            devs.assign((size_t)emu, cur);
        } else
            for (int d = 0; d < ndev; ++d) devs.push_back(d);
        if (devs.size() == 1) return create_one(prm, devs[0], precision);
        const int G = (int)devs.size();
        GroupShared *sh = new GroupShared();
        sh->G = G; sh->dev = devs;
        sh->next_ptr.assign((size_t)G, nullptr); sh->fall_ptr.assign((size_t)G, nullptr);
        sh->ev.assign((size_t)G * 2, nullptr);
        dmpc_ctx *root = nullptr;
        bool ok = true;
        for (int r = 0; r < G && ok; ++r) {
            dmpc_ctx *c = create_one(prm, devs[(size_t)r], precision);
            if (!c) { ok = false; break; }
            c->grp = sh; c->nranks = G; c->rank = r; c->grp_emulated = emu > 0;
            if (r == 0) root = c; else root->peers.push_back(c);
            for (int u = 0; u < 2 && ok; ++u) ok = hipEventCreateWithFlags(&sh->ev[(size_t)r * 2 + u], hipEventDisableTiming) == hipSuccess;
            // direct loads / stores and copies between the GPUs of the group over xGMI
            for (int q = 0; q < G && ok; ++q)
                if (devs[(size_t)q] != devs[(size_t)r]) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(devs[(size_t)q], 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // copies still work (staged)
                }
        }
        if (!ok) {
            if (g_err.empty()) g_err = "dmpc_create: group initialisation failed";
            if (root) dmpc_destroy(root); else delete sh;
            return nullptr;
        }
        (void)hipSetDevice(devs[0]);
        return root;
    }
    if (device < 0 || device >= ndev) { g_err = "dmpc_create: device index out of range"; return nullptr; }
    return create_one(prm, device, precision);
}

extern "C" int dmpc_group_size(const dmpc_ctx *ctx) { return ctx ? (ctx->grp ? ctx->grp->G : 1) : 0; }

extern "C" void dmpc_destroy(dmpc_ctx *ctx)
{
    if (!ctx) return;
    for (dmpc_ctx *pc : ctx->peers) dmpc_destroy(pc);
    ctx->peers.clear();
    for (dmpc_ctx *ch : ctx->children) dmpc_destroy(ch);
    ctx->children.clear();
    if (ctx->comm) (void)dmpc_comm_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->grp && ctx->rank == 0) {   // the root goes last: the shared block with it
        for (hipEvent_t ev : ctx->grp->ev) if (ev) (void)hipEventDestroy(ev);
        delete ctx->grp;
    }
    ctx->grp = nullptr;
    for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.t0); (void)hipEventDestroy(ev.t1); (void)hipEventDestroy(ev.t2); }
    for (auto &ev : ctx->ev_pool) { (void)hipEventDestroy(ev.t0); (void)hipEventDestroy(ev.t1); (void)hipEventDestroy(ev.t2); }
    for (int u = 0; u < 2; ++u) if (ctx->flag_ev[u]) (void)hipEventDestroy(ctx->flag_ev[u]);
    if (ctx->flags_host) (void)hipHostFree(ctx->flags_host);
    if (ctx->dbg) (void)hipFree(ctx->dbg);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    hipStream_t st = ctx->stream;
    delete ctx;   // releases every DevBuf
    if (st) (void)hipStreamDestroy(st);
}

extern "C" int dmpc_set_params(dmpc_ctx *ctx, const dmpc_params *prm)
{
    if (!ctx) { g_err = "dmpc_set_params: ctx is NULL"; return -1; }
    std::string why;
    if (check_params(prm, why)) FAIL(ctx, "dmpc_set_params: " + why);
    for (dmpc_ctx *pc : ctx->peers)
        if (dmpc_set_params(pc, prm)) FAIL(ctx, pc->err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->prm = *prm;
    return upload_tables(ctx);
}

extern "C" int64_t dmpc_solve_count(const dmpc_ctx *ctx)
{
    if (!ctx) return 0;
    int64_t n = ctx->solves;
    for (const dmpc_ctx *ch : ctx->children) n += ch->solves;
    for (const dmpc_ctx *pc : ctx->peers) n += dmpc_solve_count(pc);
    return n;
}

extern "C" int dmpc_profile(dmpc_ctx *ctx, int enable)
{
    if (!ctx) return -1;
    ctx->profile = enable;
    if (enable && ctx->ev_pool.size() < 64) {   // a pool for the next steps, created outside any timed loop
        HIPCHK(ctx, hipSetDevice(ctx->device));
        while (ctx->ev_pool.size() < 64) {
            dmpc_ctx::Ev ev{nullptr, nullptr, nullptr};
            HIPCHK(ctx, hipEventCreate(&ev.t0)); HIPCHK(ctx, hipEventCreate(&ev.t1)); HIPCHK(ctx, hipEventCreate(&ev.t2));
            ctx->ev_pool.push_back(ev);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------

static bool variant_soft(int v)
{
    return v == DMPC_VAR_BOUND || v == DMPC_VAR_BOUND2 || v == DMPC_VAR_ALL3 || v == DMPC_VAR_SOFTALL || v == DMPC_VAR_REPAIR ||
           v == DMPC_VAR_CPP || v == DMPC_VAR_CPP2 || v == DMPC_VAR_CPP1 || v == DMPC_VAR_SOFTALL_C;
}

// row capacity per agent.  Rows live in global scratch (40-64 B each); LDS only holds 4-12 B per row
// (working-set flags, slack value), so the exact worst case is affordable up to a few thousand rows.
// The kernel flags DMPC_ST_CAPACITY if a cap is ever exceeded (never silently truncated).
static int row_capacity(int variant, int N)
{
    const long nb = N > 1 ? N - 1 : 1;
    long want, cap;
    switch (variant) {
    case DMPC_VAR_HARD: want = (long)K * nb; cap = 640; break;
    case DMPC_VAR_SCP: want = (long)K * nb; cap = 4096; break;     // every neighbour at every step of addConstr (up to all k_hor of them), after exact pruning     // every k, neighbours with d < 1 (CollConstrHardDMPC.m:19), after exact pruning
    case DMPC_VAR_ALL3: want = 3 * nb; cap = 384; break;           // three steps x neighbours with d < 3 rmin
    case DMPC_VAR_BOUND: case DMPC_VAR_BOUND2: case DMPC_VAR_ONDEMAND: case DMPC_VAR_CPP: case DMPC_VAR_CPP2: want = nb; cap = 128; break;   // d < 3 rmin only
    default: want = nb; cap = 4096; break;                         // ellip / softall / repair: all N-1 neighbours
    }
    long r = want < cap ? want : cap;
    if (r < 8) r = 8;
    return (int)((r + 1) & ~1L);
}

// Working-set capacity of the first solve launch.  The capacity is a template parameter of the solve kernels: 32 / 48 / 64
// (slack-carrying variants), 48 (slack-free: 45 variables => at most 45 independent active rows, one tier).
// Slack variants, deep launches: 48 slots first -- 17 KB of LDS per wave, still 8 resident agents per CU -- and a second launch
// with 64 for the agents that outgrow them (none at N = 100; 4 in 10^4 at N = 10^4).  Round 1 used 32 slots first: the agents
// that outgrow 32 are exactly the long ones (several retry-ladder levels, many active rows) and re-solving them in a second,
// serialized launch cost more than anything else in the step.  Measured (bench secondaries, 51 200 agents of
// solveSoftDMPCbound): 32/64 tiers 1.38 ms per step, one 64-slot tier (5 agents per CU) 1.07 ms, 48/64 tiers 1.02 ms;
// 512 whole transitions 110 / 107 / 100 ms.  development option tier1_qcap = 32 | 64 selects the other forms (tests cover the 32/64 hand-off).
static int tier1_qcap(const dmpc_ctx *ctx, int variant, int scene_agents);
// hard: 45 variables => at most 45 independent active rows; 48 leaves room for a numerically near-dependent addition
static int full_qcap(int variant) { return variant_soft(variant) ? QMAX : 48; }

// what follows a solve in a closed loop (post_step_kernel); launch_step folds it into the solve kernel when the launch is tiny and
// single-tier, and reports that in ctx->post_fused
struct PostStep {
    int KT, k;
    double tol;
    double *xp, *xv, *xa, *pk, *vk, *ak;
    int *flags, *done;
};

static int tier1_qcap(const dmpc_ctx *ctx, int variant, int scene_agents)
{
    if (!variant_soft(variant)) return 48;
    if (ctx->tier1_env == 32 || ctx->tier1_env == 48 || ctx->tier1_env == 56 || ctx->tier1_env == 64) return ctx->tier1_env;
    // Large scenes: 56 slots first.  Far from its goal an agent saturates most of its 45 acceleration bounds (the crash start appends up
    // to 44 of them), and with a handful of rows and their pins the working set peaks at 48-50 slots: at N = 10^4 (C4) 50-75 agents per
    // step outgrew a 48-slot tier, none needs more than 50 -- and the few that overflow are re-solved from scratch in a second,
    // serialized launch that lasts as long as its slowest agent (0.56 ms of a 2.7 ms step).  56 slots cost 3.6 KB of LDS per agent
    // (5 instead of 6 one-agent workgroups per CU) and take them all: solve 1.81 -> 1.44 ms per step.
    // (round 5: solveSoftDMPCall too, at any scene size -- its agents carry three rows per neighbour, 2-3 % of them outgrow 48 slots, and the second
    // launch that re-solves those from scratch lasted 1.6 ms of a 4.5 ms step of 512 scenes: 3.16 -> 2.17 ms of solve launches per step)
    return (scene_agents >= 1024 || variant == DMPC_VAR_ALL3) ? 56 : 48;
}

static int launch_step(dmpc_ctx *ctx, int S, int G, int C, int g_local, int c_first, int c_count, const double *lT,
                       const double *x_p, const double *x_v, const double *x_a, const double *pf, double *p_out,
                       double *v_out, double *a_out, double *lT_next, int32_t *status, int32_t *info, hipStream_t st,
                       const int *scene_done = nullptr, int short_from = 0, const float *lTf = nullptr, const PostStep *post = nullptr,
                       const double *own_prev = nullptr /* mixed: fp64 predictions of chunk g_local [S][3K][C] when lT is not the full fp64 table */)
{
    const dmpc_params &p = ctx->prm;
    const bool soft = variant_soft(p.variant);
    StepParams P;
    memset(&P, 0, sizeof(P));
    P.variant = p.variant; P.S = S; P.G = G; P.C = C; P.g_local = g_local;
    P.c_first = c_first; P.c_count = c_count;
    P.nrmax = row_capacity(p.variant, G * C);
    if (p.variant == DMPC_VAR_HARD && (G > 256 || C >= (1 << 20)))   // packing of the scan's candidate list
        FAIL(ctx, "solveHardDMPC scan: at most 256 chunks of fewer than 2^20 agents");
    P.max_tries = p.max_tries;
    P.ell_order = p.order;
    P.h = p.h; P.rmin = p.rmin; P.e1z = 1.0 / p.c; P.e2z = p.order == 4 ? 1.0 / (p.c * p.c * p.c * p.c) : 1.0 / (p.c * p.c);   // E1 = E^-1, E2 = E^-order
    if (p.variant == DMPC_VAR_SCP) { P.e1z = 1.0; P.e2z = 1.0; }   // solveDMPC: plain Euclidean norm (CheckCollDMPC.m:6, CollConstrDMPC.m:12-13)
    P.alim = p.alim; P.Q1 = p.Q1; P.S1 = p.S1; P.term = p.term;
    P.Qfar = p.Qfar > 0 ? p.Qfar : 1000.0; P.Qnear = p.Qnear > 0 ? p.Qnear : 10000.0; P.Sfree = p.Sfree > 0 ? p.Sfree : 10.0;
    for (int d = 0; d < 3; ++d) { P.pmin[d] = p.pmin[d]; P.pmax[d] = p.pmax[d]; }
    P.tables = ctx->d_tables;
    for (int i = 0; i < 3; ++i) P.hsum[i] = ctx->hsum[i];
    // mixed precision: the scan reads the fp32 copy lTf of the table; lT (fp64, chunk g_local) is the solve's fallback
    P.lT = lTf ? (const double *)lTf : lT; P.own_prev = lTf ? (own_prev ? own_prev : lT + (size_t)g_local * S * N3 * C) : nullptr; P.x_p = x_p; P.x_v = x_v; P.x_a = x_a; P.pf = pf;
    P.p_out = p_out; P.v_out = v_out; P.a_out = a_out; P.lT_next = lT_next;
    P.status = status; P.info = info;
    {
        const size_t agents = (size_t)S * c_count;
        if (ctx->rowbuf.ensure(agents * P.nrmax * (soft ? 7 : 4) * 8) || ctx->rowkc.ensure(agents * P.nrmax * 4) ||
            ctx->hdr.ensure(agents * 8 * 4) || ctx->order.ensure(agents * 4) || ctx->counter.ensure(16) || ctx->flag_list.ensure(agents * 4))
            FAIL(ctx, "device allocation failed (row scratch)");
        P.rowbuf = ctx->rowbuf.as<double>(); P.rowkc = ctx->rowkc.as<int>(); P.hdr = ctx->hdr.as<int>();
    }
    P.dbg = ctx->dbg; P.dbg_agent = ctx->dbg_agent; P.dbg_cap = ctx->dbg_cap;
    P.iter_cap = ctx->iter_cap;
    P.rsolve_cap = ctx->rsolve_cap;
    P.scp_tol = p.tol;
    P.dep_tol_f32 = std::pow(10.0, -(double)ctx->f32_dep_exp);
    P.no_level_check = ctx->no_level_check;
    P.no_level_skip = ctx->no_level_skip;
    // (not for solveHardDMPC: rows at every horizon step, 3 % of the agents would qualify and every scan would pay for the test)
    P.fast_exit = (ctx->no_fast_exit || p.variant == DMPC_VAR_HARD || p.variant == DMPC_VAR_SCP || p.order == 4) ? 0 : 1;
    // measured: the crash start pays for the slack-carrying variants (C4, N = 10^4: solve launch -16 %) and costs on solveHardDMPC
    // (C2: -16 % throughput: with rows at every horizon step the bounds violated at the unconstrained minimiser are a poor guess)
    P.crash_min = (soft || ctx->crash_any) ? ctx->crash_min : 0;
    P.pivot_explore = ctx->pivot_explore;
    // tiny launches (a scene or a few, every agent resident at once: bound by the latency of their slowest agent, LDS is no
    // constraint) solve with the full working-set capacity in one launch; larger ones use the first tier and re-solve the few
    // agents that outgrow it (the smaller footprint also puts 6 instead of 4 one-agent workgroups on a CU: 512 transitions
    // in two halves of 25 600 agents 75 -> 63 ms); from `shallow` up the first tier runs as persistent waves
    const long ncu = ctx->num_cu > 0 ? ctx->num_cu : 256;
    const bool tiny = (long)S * c_count < 8L * ncu && !ctx->force_persist && !ctx->tier1_env;
    // (round 4, with the fitted launch-order key: launches whose agents are HEAVY -- the all-neighbour variants in scenes of >= 200 agents, every
    // violating agent carries a row per neighbour: C3 16 x 1 000 agents 37 iterations each, C5 64 x 200 agents 29 -- are throughput-bound from a
    // quarter of that depth on: persistent waves 1.50 / 0.89 ms against 1.77 / 1.05.  Light launches of the same depth -- 128 scenes x 100 agents
    // of solveSoftDMPC at MPC step 12, one iteration per agent -- stay with one agent per workgroup: 0.21 against 0.25 ms.)
    const bool heavy_agents = (p.variant == DMPC_VAR_SOFTALL || p.variant == DMPC_VAR_SOFTALL_C || p.variant == DMPC_VAR_REPAIR || p.variant == DMPC_VAR_ELLIP || p.variant == DMPC_VAR_CPP1) && G * C >= 200;
    // (crossover, agents per launch: C3 4 000: 0.72 / 0.71 ms, 8 000: 1.06 / 0.96; C5 3 200: 0.37 / 0.47, 6 400: 0.59 / 0.61 -- one agent per workgroup / persistent)
    // (round 5: the slack variants in LARGE scenes -- the 56-slot tier, agents of ~100 us each -- are bound by their work per wave slot: persistent
    // waves with the split factor, seven per CU, from two launches' worth of one-agent workgroups on)
    const bool f32t = (ctx->precision & DMPC_PREC_F32FACTOR) != 0 && p.variant != DMPC_VAR_ALL3;   // (solveSoftDMPCall keeps the fp64 factor under every precision)
    const bool big_soft = soft && G * C >= 1024 && !ctx->no_split_t && !f32t && !ctx->single_tier && !ctx->tier1_env;
    const bool shallow = (long)S * c_count < (big_soft ? 8L : (heavy_agents ? 28L : 128L)) * ncu && !ctx->force_persist && !ctx->tier1_env;
    // fp32 inverse factor: one tier with the full capacity, no split T.  Not for solveSoftDMPCall: its three nearly parallel rows per neighbour
    // need the fp64 factor (sweep of round 4: 1 % of its agent-steps ended on another ladder level) -- that variant keeps it whatever the context says.
    const int q1 = (ctx->single_tier || tiny || f32t) ? full_qcap(p.variant) : tier1_qcap(ctx, p.variant, G * C), q2 = full_qcap(p.variant);
    const bool two_tier = q1 < q2;
    ctx->post_fused = 0;
    if (post && tiny && !two_tier && g_local == 0 && G == 1 && !ctx->no_fuse && p.variant != DMPC_VAR_SCP) {
        if (ctx->post_acc.ensure((size_t)S * 16 + 64)) FAIL(ctx, "device allocation failed (post-step accumulators)");
        if (ctx->post_acc_S != S) {   // zero once per batch shape; the last wave of a scene leaves them zeroed again
            HIPCHK(ctx, hipMemsetAsync(ctx->post_acc.p, 0, (size_t)S * 16 + 64, st));
            ctx->post_acc_S = S;
        }
        P.post_on = 1; P.post_KT = post->KT; P.post_k = post->k; P.post_tol = post->tol;
        P.post_xp = post->xp; P.post_xv = post->xv; P.post_xa = post->xa; P.post_pk = post->pk; P.post_vk = post->vk; P.post_ak = post->ak;
        P.post_flags = post->flags; P.post_done = post->done;
        P.post_max = ctx->post_acc.as<unsigned long long>();
        P.post_or = (int *)(ctx->post_acc.as<unsigned long long>() + S); P.post_cnt = P.post_or + S;
        ctx->post_fused = 1;
    }
    P.scene_done = scene_done;
    P.short_from = short_from;   // unequal clusters: chunks from here on hold C-1 agents (dmpc_multigpu.hip)
    const size_t lds0 = scan_lds_bytes();
    const size_t lds1 = solve_lds_bytes(P.nrmax, soft, q1, false, 0, f32t) + (size_t)ctx->lds_pad_kb * 1024, lds2 = solve_lds_bytes(P.nrmax, soft, q2, false, 0, f32t) + (size_t)ctx->lds_pad_kb * 1024;
    const size_t ldsmax = lds2 > lds1 ? lds2 : lds1;
    if (ctx->lds_pad_kb < 0 || ldsmax > 160 * 1024) FAIL(ctx, "development option lds_pad_kb: the solve workgroup's LDS block would exceed the CU's 160 KB");
    if ((int)ldsmax > ctx->max_lds_set) {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<true, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<true, 56>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<false, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<true, 64, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_kernel<false, 48, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax));
        ctx->max_lds_set = (int)ldsmax;
    }
    dmpc_ctx::Ev ev{nullptr, nullptr, nullptr};
    if (ctx->profile) {   // event triples are recycled (dmpc_profile_read2 returns them to the pool): no event is created inside a timed loop
        if (!ctx->ev_pool.empty()) { ev = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else {
            HIPCHK(ctx, hipEventCreate(&ev.t0));
            HIPCHK(ctx, hipEventCreate(&ev.t1));
            HIPCHK(ctx, hipEventCreate(&ev.t2));
        }
        HIPCHK(ctx, hipEventRecord(ev.t0, st));
    }
    const dim3 grid((unsigned)(S * c_count)), block(64);
    // neighbour culling boxes (worth it once a scene has more than a few chunks of neighbours)
    // only for the variants whose scan and rows have a finite neighbour radius (d < 1 for the hard rows, d < 3 rmin for the
    // near-neighbour selections); solveEllipDMPC / solveSoftDMPC / solveSoftDMPCrepair take every neighbour
    const bool finite_radius = p.variant == DMPC_VAR_HARD || p.variant == DMPC_VAR_BOUND || p.variant == DMPC_VAR_BOUND2 ||
                               p.variant == DMPC_VAR_ALL3 || p.variant == DMPC_VAR_ONDEMAND || p.variant == DMPC_VAR_CPP ||
                               p.variant == DMPC_VAR_CPP2;
    if (G * C >= ctx->cull_min && !ctx->no_cull && finite_radius) {
        const int total = G * S * C;
        if (ctx->bbox.ensure((size_t)total * 6 * NSEG * 4) || ctx->bbox_nm.ensure((size_t)total * NBOX_NM * 4)) FAIL(ctx, "device allocation failed (bbox)");
        if (C >= (1 << 20) || G > 2047)   // a list entry packs (chunk << 20) | column into an int
            FAIL(ctx, "neighbour lists: at most 2047 chunks of fewer than 2^20 agents");
        // neighbour lists from the boxes (nbr_kernel): up to 4096 entries per agent, within 1 GB of scratch
        const size_t agents = (size_t)S * c_count;
        long cap = ((long)G * C + 63) & ~63L;
        if (cap > 4096) cap = 4096;
        while (cap > 256 && agents * (size_t)cap * 4 > ((size_t)1 << 30)) cap >>= 1;
        if (ctx->nbr_list.ensure(agents * (size_t)cap * 4) || ctx->nbr_cnt.ensure(agents * 4 * NBR_PARTS)) FAIL(ctx, "device allocation failed (neighbour lists)");
        const double Rsel = (p.variant == DMPC_VAR_HARD) ? 1.0 : 3.0 * p.rmin;
        const double R = Rsel * 1.0001 + 1e-4;   // a little more than the scan's radius: conservative in fp32 too
        // round 4: lists from a cell grid, filtered by the fp32 distance test (grid_query_kernel); the all-pairs box test of round 3 stays
        // behind option nbr_grid = 0 (A/B runs, tests) and for scenes whose bitmap would not fit a wave's LDS
        const size_t gq_lds = grid_query_lds(G * C);
        // (from grid_min agents per scene on: in a scene of a few hundred agents the reach of a query covers most of the workspace and the
        // all-pairs test with the neighbours' boxes as scalar operands is the cheaper pass -- tools/gpu_grid_min_ab.py, 102 400 agents, scan
        // side all-pairs / grid: hard rows 400 agents per scene 0.81 / 0.81 ms, 800: 1.07 / 0.91, 1 600: 1.42 / 1.06, 3 200: 1.92 / 1.24;
        // solveSoftDMPCbound 400: 0.62 / 0.66, 800: 0.71 / 0.69, 1 600: 0.84 / 0.73, 3 200: 1.04 / 0.81.  A rank that queries ONE chunk of
        // 8 x 100 agents per scene still bins all 800: 0.87 against 0.64 ms, `bench.py --emulate-gpus 8 --debug-option grid_min=512`)
        const int grid_from = (c_count == G * C) ? ctx->grid_min : ctx->grid_min_part;
        const bool use_grid = ctx->nbr_grid && G * C >= grid_from && gq_lds <= 64 * 1024;
        // (grid geometry and buffer first: the counters are zeroed by the neighbour-major copy kernel, which runs anyway -- a memset of an odd
        // size is two fill launches, 9 us)
        GridGeom gg{};
        int ncell = 1;
        int *g_cnt = nullptr, *g_mh = nullptr, *g_st = nullptr, *g_cell = nullptr, *g_pos = nullptr;
        f4_t *g_ent = nullptr;
        size_t n_zero = 0;
        bool fused = false;   // ONE scene: the grid in two launches (grid_prep_kernel, grid_fill2_kernel) instead of five
        if (use_grid) {
            // cells: R along x (the cells of a run along x are contiguous in the entry array: their granularity is free), 1.5 R along y
            // and 1.5 R c along z (the metric's z scale), at most 32 per axis
            const double cell[3] = {R, 1.5 * R, 1.5 * R * p.c};
            for (int a = 0; a < 3; ++a) {
                const double span = p.pmax[a] - p.pmin[a];
                int n = (int)(span / cell[a]);
                n = n < 1 ? 1 : (n > 32 ? 32 : n);
                gg.n[a] = n; gg.org[a] = (float)p.pmin[a]; gg.inv[a] = (float)(n / (span > 0 ? span : 1.0));
                ncell *= n;
            }
            // one grid per third of the horizon (keyed by the centre of that segment's box: a third of the extent of the whole horizon's).  One
            // buffer: [S][3][ncell] counts, [S][3][3] largest half extents (zeroed together), [S][3][ncell + 1] starts, [3][G S C] cells, [S][3][G C] entries
            const size_t n_cnt = (size_t)S * NSEG * ncell, n_mh = (size_t)S * NSEG * 3, n_st = (size_t)S * NSEG * (ncell + 1);
            const size_t n_hd = (n_cnt + n_mh + n_st + 2 * (size_t)NSEG * total + 7) & ~(size_t)7;   // (the entry records behind it are 32-byte aligned)
            if (ctx->grid.ensure((n_hd + 8 * (size_t)NSEG * total) * 4)) FAIL(ctx, "device allocation failed (neighbour grid)");
            g_cnt = ctx->grid.as<int>(); g_mh = g_cnt + n_cnt; g_st = g_mh + n_mh; g_cell = g_st + n_st; g_pos = g_cell + (size_t)NSEG * total;
            g_ent = (f4_t *)(g_cnt + n_hd);
            n_zero = n_cnt + n_mh;
            fused = S == 1 && ctx->prep_fuse && (size_t)NSEG * (ncell + 1) * 4 <= 48 * 1024;
        }
        if (!fused) {
            if (lTf) hipLaunchKernelGGL(bbox_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, C, lTf, ctx->bbox.as<float>(), ctx->bbox_nm.as<float>());
            else hipLaunchKernelGGL(bbox_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, C, lT, ctx->bbox.as<float>(), ctx->bbox_nm.as<float>());
        }
        if (p.variant != DMPC_VAR_HARD || use_grid) {   // neighbour-major fp32 copy of the table: the list walk of the per-step distance scan, the distance test of the grid query
            const size_t tot = (size_t)total * 64;
            if (ctx->lrow.ensure(tot * 4)) FAIL(ctx, "device allocation failed (neighbour-major table)");
            if (fused) {}   // (grid_prep_kernel below makes the copy)
            else if (lTf) hipLaunchKernelGGL(table_nbrmajor_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tot, C, lTf, ctx->lrow.as<float>(), g_cnt, n_zero);
            else hipLaunchKernelGGL(table_nbrmajor_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tot, C, lT, ctx->lrow.as<float>(), g_cnt, n_zero);
            if (p.variant != DMPC_VAR_HARD) P.lrow = ctx->lrow.p;
        }
        if (use_grid && fused) {
            // the counters are zero when the last scan launch left them so (for this buffer and size); a memset otherwise (first step, another batch shape in between)
            const unsigned long long key = (unsigned long long)(size_t)g_cnt ^ ((unsigned long long)n_zero << 48) ^ ((unsigned long long)total << 20);
            if (!ctx->grid_clean || ctx->grid_clean_key != key) HIPCHK(ctx, hipMemsetAsync(g_cnt, 0, n_zero * 4, st));
            ctx->grid_clean = false; ctx->grid_clean_key = key;
            P.gzero = g_cnt; P.gzero_n = (int)n_zero;
            const int nbA = (total + 255) / 256, nbC = (int)(((size_t)total * 64 + 255) / 256);
            if (lTf) hipLaunchKernelGGL(grid_prep_kernel<float>, dim3((unsigned)(nbA + nbC)), dim3(256), 0, st, total, C, short_from, gg, nbA, lTf, ctx->bbox.as<float>(), ctx->bbox_nm.as<float>(), ctx->lrow.as<float>(), g_cell, g_pos, g_cnt, g_mh);
            else hipLaunchKernelGGL(grid_prep_kernel<double>, dim3((unsigned)(nbA + nbC)), dim3(256), 0, st, total, C, short_from, gg, nbA, lT, ctx->bbox.as<float>(), ctx->bbox_nm.as<float>(), ctx->lrow.as<float>(), g_cell, g_pos, g_cnt, g_mh);
            hipLaunchKernelGGL(grid_fill2_kernel, dim3((unsigned)nbA), dim3(256), (size_t)NSEG * (ncell + 1) * 4, st, total, C, ncell, (float)(1.0 / p.c), (const int *)g_cell, (const int *)g_pos, (const int *)g_cnt, g_st, (const float *)ctx->lrow.as<float>(), g_ent);
        } else if (use_grid) {
            ctx->grid_clean = false;
            hipLaunchKernelGGL(grid_bin_kernel, dim3((unsigned)((total + 255) / 256), NSEG), dim3(256), 0, st, total, S, C, short_from, gg, (const float *)ctx->bbox_nm.as<float>(), g_cell, g_cnt, g_mh);
            hipLaunchKernelGGL(grid_scan_kernel, dim3((unsigned)(S * NSEG)), dim3(ncell > 512 ? 1024 : 256), 0, st, ncell, g_cnt, g_st);
            hipLaunchKernelGGL(grid_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, S, C, ncell, (float)(1.0 / p.c), (const int *)g_cell, g_cnt, (const int *)g_st, (const float *)ctx->lrow.as<float>(), g_ent);
        }
        if (use_grid) {
            const int nq = S * c_count;
            hipLaunchKernelGGL(grid_query_kernel, dim3((unsigned)nq), dim3(64 * GQ_WAVES), gq_lds, st, S, G, C, g_local, c_first, c_count, gg,
                               (float)R, (float)(R * p.c), (float)(1.0 / p.c), (float)(Rsel * Rsel * 1.002), (const float *)ctx->bbox_nm.as<float>(), (const float *)ctx->lrow.as<float>(),
                               (const int *)g_st, (const f4_t *)g_ent, (const int *)g_mh, (int)cap, (G == 1 && c_first == 0 && c_count == C && !short_from) ? 1 : 0,
                               ctx->nbr_list.as<int>(), ctx->nbr_cnt.as<int>());
        } else {
            const int nblk = (c_count + 63) / 64;
            hipLaunchKernelGGL(nbr_kernel, dim3((unsigned)(S * nblk * NBR_PARTS)), dim3(64), 0, st, S, G, C, g_local, c_first, c_count, short_from, (float)R, (float)(R * p.c),
                               (const float *)ctx->bbox.as<float>(), (const float *)ctx->bbox_nm.as<float>(), (int)cap, ctx->nbr_list.as<int>(), ctx->nbr_cnt.as<int>());
        }
        P.nbr_cap = (int)cap; P.nbr_list = ctx->nbr_list.as<int>(); P.nbr_cnt = ctx->nbr_cnt.as<int>();
    }
    if (p.variant == DMPC_VAR_SCP) {
        // solveDMPC.m: the whole SCP loop of an agent -- up to k_hor passes of {scan about the previous pass's prediction, slack-free QP} -- in ONE
        // launch, one agent per 64-thread workgroup (dmpc_scp_kernel); no neighbour lists (rows for every other agent), no launch order
        if (lTf) FAIL(ctx, "solveDMPC (DMPC_VAR_SCP) runs in fp64 only: create the context with DMPC_PREC_F64");
        P.qcap = 48; P.only_flagged = 0; P.qover_bit = ST_CAPACITY; P.lds_per_wave = (int)lds0;
        const size_t lds_scp = solve_lds_bytes(P.nrmax, false, 48, false) > lds0 ? solve_lds_bytes(P.nrmax, false, 48, false) : lds0;
        if ((int)lds_scp > ctx->max_lds_scp) {
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_scp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scp));
            ctx->max_lds_scp = (int)lds_scp;
        }
        if (ctx->profile) HIPCHK(ctx, hipEventRecord(ev.t1, st));
        hipLaunchKernelGGL(dmpc_scp_kernel, grid, block, lds_scp, st, P);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->profile) {
            HIPCHK(ctx, hipEventRecord(ev.t2, st));
            ctx->events.push_back(ev);
        }
        ctx->solves += (int64_t)S * c_count;
        return 0;
    }
    const bool run_order = ctx->forced_n != S * c_count && S * c_count >= 512 && !ctx->no_lpt;
    // the reduced solver (dmpc_rsolve.hip) takes solveSoftDMPCbound / bound2 and DMPC::solveQPv2 in every launch form: which kernel solves an agent must not depend on how deep the launch is
    const bool reduced = ctx->reduced_solver && (p.variant == DMPC_VAR_BOUND || p.variant == DMPC_VAR_BOUND2 || p.variant == DMPC_VAR_CPP || p.variant == DMPC_VAR_CPP2) && !f32t && ctx->num_cu >= 1;   // the variants with slack rows on ONE horizon step
    P.zero4 = (!tiny || run_order || reduced) ? ctx->counter.as<int>() : nullptr;   // queue heads of the persistent solve launches, tier-2 count, live bound: zeroed by the scan kernel (a memset is a launch of its own, 5 us)
    // phase 0: scan + rows
    P.qcap = q1; P.only_flagged = 0; P.qover_bit = two_tier ? ST_QOVER : ST_CAPACITY;
    {
        // several independent waves per workgroup (fewer workgroups to dispatch), as many as fit the default 64 KB of
        // dynamic LDS (the neighbour list of large scenes can take 37 KB per wave)
        const int total = S * c_count;
        int W = SCAN_WAVES_PER_WG;
        while (W > 1 && lds0 * W > 64 * 1024) W >>= 1;
        const dim3 sgrid((unsigned)((total + W - 1) / W)), sblock(64u * W);
        P.lds_per_wave = (int)lds0;
        const bool fx = P.fast_exit != 0;
        if (p.order == 4) {   // super-ellipsoid of order 4 (all-neighbour variants): its own scan kernels, without the unconstrained exit
            if (lTf) { if (soft) hipLaunchKernelGGL((dmpc_scan_kernel<true, float, false, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<false, float, false, true>), sgrid, sblock, lds0 * W, st, P); }
            else { if (soft) hipLaunchKernelGGL((dmpc_scan_kernel<true, double, false, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<false, double, false, true>), sgrid, sblock, lds0 * W, st, P); }
        } else if (lTf) {
            if (soft) { if (fx) hipLaunchKernelGGL((dmpc_scan_kernel<true, float, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<true, float, false>), sgrid, sblock, lds0 * W, st, P); }
            else { if (fx) hipLaunchKernelGGL((dmpc_scan_kernel<false, float, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<false, float, false>), sgrid, sblock, lds0 * W, st, P); }
        } else {
            if (soft) { if (fx) hipLaunchKernelGGL((dmpc_scan_kernel<true, double, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<true, double, false>), sgrid, sblock, lds0 * W, st, P); }
            else { if (fx) hipLaunchKernelGGL((dmpc_scan_kernel<false, double, true>), sgrid, sblock, lds0 * W, st, P); else hipLaunchKernelGGL((dmpc_scan_kernel<false, double, false>), sgrid, sblock, lds0 * W, st, P); }
        }
    }
    if (P.gzero) ctx->grid_clean = true;   // (this scan launch leaves the cell grid's counters zero for the next step's grid_prep_kernel)
    // heaviest-first launch order for the solve phase (key left by the scan in hdr[7]).  Tiny launches do not need it.
    if (ctx->forced_n == S * c_count) P.order = ctx->forced_order.as<int>();   // development aid: externally supplied launch order
    else if (run_order) {
        // (slices: the kernel is a chain of dependent memory round trips per thread -- 8 workgroups of 1024 threads took 20 us for 51 200
        // agents, six agents per thread one after the other; with one agent per thread 7 us: headline 52.3 -> 53.2 M solves/s)
        const int total = S * c_count;
        int nb = ctx->order_slices > 0 ? ctx->order_slices : (total >= 65536 ? 64 : (total >= 1024 ? total / 1024 : 1));
        if ((total + nb - 1) / nb > 24576) nb = (total + 24575) / 24576;   // (a slice's keys live in LDS, 2 bytes each next to the histograms: at most 48 KB of them)
        int *hint = nullptr;
        if (ctx->order_hint) {   // the previous step's work estimates: valid while the batch keeps its shape
            if (ctx->prev_cost.ensure((size_t)total * 4)) FAIL(ctx, "device allocation failed (order hint)");
            const long shape = ((long)S << 32) ^ ((long)c_count << 8) ^ (long)p.variant;
            if (shape != ctx->prev_cost_shape) { HIPCHK(ctx, hipMemsetAsync(ctx->prev_cost.p, 0, (size_t)total * 4, st)); ctx->prev_cost_shape = shape; }
            hint = ctx->prev_cost.as<int>();
            P.cost_out = hint;
        }
        hipLaunchKernelGGL(order_kernel, dim3((unsigned)nb), dim3(1024), (size_t)((total + nb - 1) / nb) * 2, st, total, (const int *)P.hdr, ctx->order.as<int>(), ctx->counter.as<int>() + 3,
                           hint, ctx->order_hint);
        P.order = ctx->order.as<int>();
        P.live_bound = ctx->counter.as<int>() + 3;
    }
    if (ctx->profile) HIPCHK(ctx, hipEventRecord(ev.t1, st));
    // phase 1: persistent waves (one workgroup per CU, shared tables, agents claimed from a queue) when at least two
    // waves fit next to the shared tables; otherwise one agent per workgroup
    const size_t LDS_CU = 160 * 1024;
    // Slack-free variants (round 4): split T -- HARD_TS columns of the inverse factor in every wave's block, the rest of the 48 in
    // extensions that the waves of a workgroup take from a pool when an agent's working set outgrows them (dmpc_solve.hip) -- so that
    // twelve waves (three per SIMD: what 168 registers per lane allow) share a CU's LDS instead of nine.
    // Slack variants, 56-slot tier of large scenes (round 5): the same split with 48 own columns -- the eight columns beyond them (3.6 KB) come from
    // the pool for the 2 % of the agents whose working set outgrows 48 slots -- so that SEVEN waves share a CU where five one-agent workgroups
    // (30 KB each, their own copy of the tables) or six unsplit persistent waves did: the 10^4-agent scene is bound by its work per wave slot
    // (four / five resident agents per CU: 1.03 / 0.88 ms, option lds_pad_kb).
    const int tsplit_hard = (!soft && !ctx->no_split_t && !f32t) ? HARD_TS : 0;
    int tsplit = tsplit_hard;
    int n_ext = 0;
    auto persist_waves = [&](int qcap, size_t &per) -> int {
        tsplit = soft ? ((qcap == 56 && !ctx->no_split_t && !f32t) ? SOFT_TS : 0) : tsplit_hard;
        per = solve_lds_bytes(P.nrmax, soft, qcap, true, tsplit, f32t);
        int pw = (int)((LDS_CU - PERSIST_TABLE_BYTES) / per);
#ifdef DMPC_DEV_PW   // development builds: fewer persistent waves per CU (how much does a long agent lose to the wave it shares a SIMD with?)
        if (pw > DMPC_DEV_PW) pw = DMPC_DEV_PW;
#endif
        const int cap = soft ? 8 : ((tsplit || f32t) ? HARD_PW : 9);   // waves per workgroup the kernels are compiled for (launch bounds)
        pw = pw > cap ? cap : pw;
        if (tsplit) {   // the extensions need room too: at least a third as many as waves (3 % of the headline launch's agents need one, for 15 % of its iterations)
            const size_t eb = (size_t)ext_doubles(qcap, tsplit) * 8;
            for (;; --pw) {
                n_ext = (int)((LDS_CU - PERSIST_TABLE_BYTES - EXT_PAD_BYTES - (size_t)pw * per) / eb);
                if (n_ext > 31) n_ext = 31;
                if (pw < 2 || 3 * n_ext >= pw) break;
            }
            if (ctx->ext_cap > 0 && n_ext > ctx->ext_cap) n_ext = ctx->ext_cap;
        }
        return pw;
    };
    // the working-set capacity is a template parameter of the solve kernels
    auto launch_plain = [&](int qcap, size_t lds) {
        if (!P.only_flagged) ctx->last_kernel = std::string("dmpc_solve_kernel<") + (soft ? "true, " : "false, ") + std::to_string(f32t ? (soft ? 64 : 48) : (soft ? qcap : 48)) + (f32t ? ", float>" : ", double>");
        if (f32t) { if (soft) hipLaunchKernelGGL((dmpc_solve_kernel<true, 64, float>), grid, block, lds, st, P); else hipLaunchKernelGGL((dmpc_solve_kernel<false, 48, float>), grid, block, lds, st, P); }
        else if (soft && qcap == 32) hipLaunchKernelGGL((dmpc_solve_kernel<true, 32>), grid, block, lds, st, P);
        else if (soft && qcap == 48) hipLaunchKernelGGL((dmpc_solve_kernel<true, 48>), grid, block, lds, st, P);
        else if (soft && qcap == 56) hipLaunchKernelGGL((dmpc_solve_kernel<true, 56>), grid, block, lds, st, P);
        else if (soft) hipLaunchKernelGGL((dmpc_solve_kernel<true, 64>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((dmpc_solve_kernel<false, 48>), grid, block, lds, st, P);
    };
    auto launch_persist = [&](int qcap, dim3 g, dim3 b, size_t lds) {
        if (!P.only_flagged) {
            const int qc = f32t ? (soft ? 64 : 48) : (soft ? qcap : 48);
            const int ts = f32t ? qc : (tsplit ? (soft ? SOFT_TS : HARD_TS) : qc);
            ctx->last_kernel = std::string("dmpc_solve_persist_kernel<") + (soft ? "true, " : "false, ") + std::to_string(qc) + ", " + std::to_string(ts) + (f32t ? ", float>" : ", double>");
        }
        if (f32t) { if (soft) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 64, 64, float>), g, b, lds, st, P); else hipLaunchKernelGGL((dmpc_solve_persist_kernel<false, 48, 48, float>), g, b, lds, st, P); }
        else if (soft && qcap == 32) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 32>), g, b, lds, st, P);
        else if (soft && qcap == 48) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 48>), g, b, lds, st, P);
        else if (soft && qcap == 56 && tsplit) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 56, SOFT_TS>), g, b, lds, st, P);
        else if (soft && qcap == 56) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 56>), g, b, lds, st, P);
        else if (soft) hipLaunchKernelGGL((dmpc_solve_persist_kernel<true, 64>), g, b, lds, st, P);
        else if (tsplit) hipLaunchKernelGGL((dmpc_solve_persist_kernel<false, 48, HARD_TS>), g, b, lds, st, P);
        else hipLaunchKernelGGL((dmpc_solve_persist_kernel<false, 48>), g, b, lds, st, P);
    };
    auto solve_launch = [&](int qcap, size_t lds_plain, int tier, bool want_persist) -> int {
        size_t per = 0;
        const int pw = persist_waves(qcap, per);
        const int total = S * c_count;
        if (!want_persist || ctx->no_persist || pw < 2 || ctx->num_cu < 1) {
            launch_plain(qcap, lds_plain);
            return 0;
        }
        const size_t lds = PERSIST_TABLE_BYTES + (size_t)pw * per + (tsplit ? (size_t)n_ext * ext_doubles(qcap, tsplit) * 8 + EXT_PAD_BYTES : 0);
        P.n_ext = tsplit ? n_ext : 0;
        if ((int)lds > ctx->max_lds_persist) {
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 56>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 56, SOFT_TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<false, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<false, 48, HARD_TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<true, 64, 64, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(ctx, hipFuncSetAttribute((const void *)dmpc_solve_persist_kernel<false, 48, 48, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->max_lds_persist = (int)lds;
        }
        P.counter = ctx->static_queue ? nullptr : ctx->counter.as<int>() + tier;
        P.lds_per_wave = (int)per;
        int wgs = (total + pw - 1) / pw;
        if (wgs > ctx->num_cu) wgs = ctx->num_cu;
        // tickets of the queue's light bulk: single positions where a wave solves few, heavy agents (fewer than 12 per wave, or solveSoftDMPCall)
        P.queue_chunk = ctx->queue_chunk > 0 ? ctx->queue_chunk : ((total < 12 * wgs * pw || ctx->prm.variant == DMPC_VAR_ALL3) ? 1 : 2);
        launch_persist(qcap, dim3((unsigned)wgs), dim3((unsigned)(64 * pw)), lds);
        return 0;
    };
    // Measured on C2 (hard, 100 agents/scene): persistent waves win once the launch is deep enough to be
    // throughput-bound (+6 % at 102 400 agents: 8 instead of 7 resident agents per CU), while short launches are
    // bound by their single slowest agent, which runs ~4 % faster in the leaner one-agent-per-workgroup kernel.
    size_t per1 = 0, per2 = 0;
    const int pw1 = persist_waves(q1, per1), pw2 = persist_waves(q2, per2);
    const bool deep = !shallow && (big_soft || (long)S * c_count >= (heavy_agents ? 28L : 16L * (pw1 > 0 ? pw1 : 1)) * ctx->num_cu);
    // tier 2 as persistent waves over the flagged list (nearly always empty: the launch then costs a few microseconds
    // instead of one workgroup per agent just to find out that there is nothing to do)
    const bool t2_list = two_tier && !tiny && !ctx->no_persist && pw2 >= 2 && ctx->num_cu >= 1;
    if (reduced) {
        // tier 0: the reduced solver over every agent of the launch (persistent waves, as many workgroups per CU as its registers allow); the agents it
        // does not take -- more than 64 rows, a third active wall, more than five hard constraints -- go to the general solver with its full capacity
        if (ctx->rsolve_blocks == 0) {
            int nb = 0;
            HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)dmpc_rsolve_persist_kernel, RSOLVE_WAVES * 64, (size_t)RSOLVE_WAVES * RSOLVE_LDS_PER_WAVE));
            ctx->rsolve_blocks = nb > 0 ? nb : 1;
        }
        const int total = S * c_count;
        P.qcap = 0; P.only_flagged = 0; P.qover_bit = ST_QOVER;
        if (P.post_on) P.live_bound = nullptr;   // fused post-step: the agents the scan finished are visited too (their part of the state advance and of the scene's verdict)
        P.flag_count = ctx->counter.as<int>() + 2; P.flag_list = ctx->flag_list.as<int>();
        P.counter = ctx->static_queue ? nullptr : ctx->counter.as<int>();
        P.lds_per_wave = RSOLVE_LDS_PER_WAVE;
        int wgs = (total + RSOLVE_WAVES - 1) / RSOLVE_WAVES;
        if (wgs > ctx->rsolve_blocks * ctx->num_cu) wgs = ctx->rsolve_blocks * ctx->num_cu;
        P.queue_chunk = ctx->queue_chunk > 0 ? ctx->queue_chunk : (total < 12 * wgs * RSOLVE_WAVES ? 1 : 2);
        ctx->last_kernel = "dmpc_rsolve_persist_kernel";
        hipLaunchKernelGGL(dmpc_rsolve_persist_kernel, dim3((unsigned)wgs), dim3(RSOLVE_WAVES * 64), (size_t)RSOLVE_WAVES * RSOLVE_LDS_PER_WAVE, st, P);
        P.qcap = q2; P.only_flagged = 1; P.qover_bit = ST_CAPACITY;
        const bool t2p = !ctx->no_persist && pw2 >= 2;   // persistent waves over the flagged list (nearly always empty); else one workgroup per agent, each looking at its agent's flag
        P.order = t2p ? ctx->flag_list.as<int>() : nullptr; P.flag_list = nullptr; P.live_bound = nullptr;
        if (solve_launch(q2, lds2, 1, t2p)) return -1;
        HIPCHK(ctx, hipGetLastError());
        if (ctx->profile) {
            HIPCHK(ctx, hipEventRecord(ev.t2, st));
            ctx->events.push_back(ev);
        }
        ctx->solves += (int64_t)S * c_count;
        return 0;
    }
    if (t2_list) { P.flag_count = ctx->counter.as<int>() + 2; P.flag_list = ctx->flag_list.as<int>(); }
    if (solve_launch(q1, lds1, 0, deep || ctx->force_persist)) return -1;
    if (two_tier) {   // tier 2: only agents flagged ST_QOVER do any work
        P.qcap = q2; P.only_flagged = 1; P.qover_bit = ST_CAPACITY;
        if (t2_list) { P.order = ctx->flag_list.as<int>(); P.flag_list = nullptr; }
        if (solve_launch(q2, lds2, 1, t2_list)) return -1;
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profile) {
        HIPCHK(ctx, hipEventRecord(ev.t2, st));
        ctx->events.push_back(ev);
    }
    ctx->solves += (int64_t)S * c_count;
    return 0;
}

// development aid (not part of the public header): trace the active-set iterations of one agent
extern "C" int dmpc_debug_trace(dmpc_ctx *ctx, int agent, int cap, double *host_out)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (host_out && ctx->dbg) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(host_out, ctx->dbg, sizeof(double) * 8 * ctx->dbg_cap, hipMemcpyDeviceToHost));
        return 0;
    }
    if (ctx->dbg) { (void)hipFree(ctx->dbg); ctx->dbg = nullptr; }
    ctx->dbg_agent = agent; ctx->dbg_cap = cap;
    if ((agent >= 0 || agent == -2 || agent == -3 || agent == -5 || agent == -7) && cap > 0) {
        HIPCHK(ctx, hipMalloc((void **)&ctx->dbg, sizeof(double) * 8 * cap));
        HIPCHK(ctx, hipMemset(ctx->dbg, 0, sizeof(double) * 8 * cap));
    }
    return 0;
}

// development aid (not part of the public header): a coalesced streaming read of `bytes` bytes at lane_bytes (8 | 16) per lane, `reps`
// launches -- the known byte count the FETCH_SIZE counter is calibrated on (tools/gpu_fetch_calib.py under rocprofv3 --pmc FETCH_SIZE)
extern "C" int dmpc_debug_read_probe(dmpc_ctx *ctx, size_t bytes, int lane_bytes, int reps)
{
    if (!ctx || (lane_bytes != 8 && lane_bytes != 16) || bytes < 4096) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void *buf = nullptr; double *sink = nullptr;
    HIPCHK(ctx, hipMalloc(&buf, bytes));
    HIPCHK(ctx, hipMalloc((void **)&sink, 64));
    HIPCHK(ctx, hipMemsetAsync(buf, 0, bytes, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        if (lane_bytes == 8) hipLaunchKernelGGL(read_probe_kernel<double>, dim3(256 * 16), dim3(256), 0, ctx->stream, bytes / 8, (const double *)buf, sink);
        else hipLaunchKernelGGL(read_probe_kernel<double2>, dim3(256 * 16), dim3(256), 0, ctx->stream, bytes / 16, (const double2 *)buf, sink);
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(buf); (void)hipFree(sink);
    return 0;
}

// development aid (not part of the public header): the scan's hand-off headers of the last step (8 ints per agent: rows, reference row count,
// violating step, status, flags, rows exist, ladder start, launch-order key)
extern "C" int dmpc_debug_read_hdr(dmpc_ctx *ctx, int *host_out, int n_agents)
{
    if (!ctx || !host_out) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((size_t)n_agents * 32 > ctx->hdr.cap) { ctx->err = "dmpc_debug_read_hdr: more agents than the last step had"; return -1; }
    HIPCHK(ctx, hipMemcpy(host_out, ctx->hdr.p, (size_t)n_agents * 32, hipMemcpyDeviceToHost));
    return 0;
}

// development aid (not part of the public header): force the solve launch order (a permutation of the S*c_count agents
// of the next launches; n = 0 returns to the built-in policy).  Used to measure what an ideal order would give.
extern "C" int dmpc_debug_set_order(dmpc_ctx *ctx, const int *host_order, int n)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->forced_n = 0;
    if (n > 0 && host_order) {
        if (ctx->forced_order.ensure(sizeof(int) * (size_t)n)) { ctx->err = "dmpc_debug_set_order: out of device memory"; return -1; }
        HIPCHK(ctx, hipMemcpy(ctx->forced_order.p, host_order, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        ctx->forced_n = n;
    }
    return 0;
}

extern "C" int dmpc_profile_read2(dmpc_ctx *ctx, double *solve_avg_ms, double *scan_avg_ms, int64_t *n_steps)
{
    if (!ctx) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto &ev : ctx->events) {
        HIPCHK(ctx, hipEventSynchronize(ev.t2));
        float ms_scan = 0.f, ms_solve = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms_scan, ev.t0, ev.t1));
        HIPCHK(ctx, hipEventElapsedTime(&ms_solve, ev.t1, ev.t2));
        ctx->prof_scan_ms_sum += ms_scan;
        ctx->prof_ms_sum += ms_solve;
        ctx->prof_n += 1;
        ctx->ev_pool.push_back(ev);
    }
    ctx->events.clear();
    if (solve_avg_ms) *solve_avg_ms = ctx->prof_n ? ctx->prof_ms_sum / (double)ctx->prof_n : 0.0;
    if (scan_avg_ms) *scan_avg_ms = ctx->prof_n ? ctx->prof_scan_ms_sum / (double)ctx->prof_n : 0.0;
    if (n_steps) *n_steps = ctx->prof_n;
    ctx->prof_ms_sum = ctx->prof_scan_ms_sum = 0.0;
    ctx->prof_n = 0;
    return 0;
}

// average duration (ms) of the step's kernels (scan + order + solve tiers) since the previous read
extern "C" int dmpc_profile_read(dmpc_ctx *ctx, double *avg_ms, int64_t *n_launches)
{
    double a = 0.0, b = 0.0;
    const int rc = dmpc_profile_read2(ctx, &a, &b, n_launches);
    if (avg_ms) *avg_ms = a + b;
    return rc;
}

static int table_f32(dmpc_ctx *ctx, const double *src, DevBuf &dst, size_t n, hipStream_t st);
static int group_transition(dmpc_ctx *root, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol, double *pk,
                            double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status);
static int group_step_batch(dmpc_ctx *root, int S, int N, const double *l, const double *x_p, const double *x_v, const double *x_a,
                            const double *pf, double *p_out, double *v_out, double *a_out, int32_t *status, int32_t *info);

extern "C" int dmpc_step_device(dmpc_ctx *ctx, int S, int G, int C, int g_local, const double *lT, const double *x_p,
                                const double *x_v, const double *x_a, const double *pf, double *p_out, double *v_out,
                                double *a_out, double *lT_next, int32_t *status, int32_t *info, void *stream)
{
    if (!ctx) { g_err = "dmpc_step_device: ctx is NULL"; return -1; }
    if (S < 1 || G < 1 || C < 1 || g_local < 0 || g_local >= G) FAIL(ctx, "dmpc_step_device: bad S/G/C/g_local");
    if (!lT || !x_p || !x_v || !x_a || !pf || !p_out || !v_out || !a_out || !status) FAIL(ctx, "dmpc_step_device: NULL pointer");
    if (ctx->grp) FAIL(ctx, "dmpc_step_device: device pointers belong to ONE GPU; a DMPC_DEVICE_ALL context drives several (use the host-pointer entry points)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // mixed precision: the caller's table stays fp64; the scan reads an fp32 copy made here (half the bytes of the O(N) part)
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;
    if (mixed && table_f32(ctx, lT, ctx->lTf, (size_t)G * S * N3 * C, (hipStream_t)stream)) return -1;
    return launch_step(ctx, S, G, C, g_local, 0, C, lT, x_p, x_v, x_a, pf, p_out, v_out, a_out, lT_next, status, info,
                       (hipStream_t)stream, nullptr, 0, mixed ? ctx->lTf.as<float>() : nullptr);
}

extern "C" int dmpc_table_from_rows_device(dmpc_ctx *ctx, int S, int G, int C, const double *rows, double *lT, void *stream)
{
    if (!ctx) { g_err = "dmpc_table_from_rows_device: ctx is NULL"; return -1; }
    if (S < 1 || G < 1 || C < 1 || !rows || !lT) FAIL(ctx, "dmpc_table_from_rows_device: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)S * G * C * N3;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(table_from_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, G, C, rows, lT);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C" int dmpc_advance_device(dmpc_ctx *ctx, int count, const double *p_out, const double *v_out, const double *a_out,
                                   const int32_t *status, double *x_p, double *x_v, double *x_a, void *stream)
{
    if (!ctx) { g_err = "dmpc_advance_device: ctx is NULL"; return -1; }
    if (count < 1 || !p_out || !v_out || !a_out || !status || !x_p || !x_v || !x_a) FAIL(ctx, "dmpc_advance_device: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(advance_kernel, dim3((unsigned)((count * 3 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, count, p_out, v_out, a_out, (const int *)status, x_p, x_v, x_a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// host-pointer entry points
// ---------------------------------------------------------------------------------------------

static int ensure_step_scratch(dmpc_ctx *ctx, size_t agents_table, size_t agents_solved)
{
    int rc = 0;
    rc |= ctx->rows.ensure(agents_table * N3 * 8);
    rc |= ctx->lT.ensure(agents_table * N3 * 8);
    rc |= ctx->xp.ensure(agents_solved * 24);
    rc |= ctx->xv.ensure(agents_solved * 24);
    rc |= ctx->xa.ensure(agents_solved * 24);
    rc |= ctx->pf.ensure(agents_solved * 24);
    rc |= ctx->pout.ensure(agents_solved * N3 * 8);
    rc |= ctx->vout.ensure(agents_solved * N3 * 8);
    rc |= ctx->aout.ensure(agents_solved * N3 * 8);
    rc |= ctx->status.ensure(agents_solved * 4);
    rc |= ctx->info.ensure(agents_solved * 32);
    if (rc) FAIL(ctx, "device allocation failed");
    return 0;
}

// mixed precision: fp32 copy of a table (n doubles) for the scan
static int table_f32(dmpc_ctx *ctx, const double *src, DevBuf &dst, size_t n, hipStream_t st)
{
    if (dst.ensure(n * 4)) FAIL(ctx, "device allocation failed (fp32 table)");
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(table_to_f32_kernel, dim3(blocks), dim3(256), 0, st, n, src, dst.as<float>());
    return 0;
}

extern "C" int dmpc_step_batch(dmpc_ctx *ctx, int S, int N, const double *l, const double *x_p, const double *x_v,
                               const double *x_a, const double *pf, double *p_out, double *v_out, double *a_out,
                               int32_t *status, int32_t *info)
{
    if (!ctx) { g_err = "dmpc_step_batch: ctx is NULL"; return -1; }
    if (S < 1 || N < 1) FAIL(ctx, "dmpc_step_batch: S and N must be >= 1");
    if (!l || !x_p || !x_v || !x_a || !pf || !p_out || !v_out || !a_out || !status) FAIL(ctx, "dmpc_step_batch: NULL pointer");
    if (ctx->grp && N >= ctx->grp->G) return group_step_batch(ctx, S, N, l, x_p, x_v, x_a, pf, p_out, v_out, a_out, status, info);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t A = (size_t)S * N;
    if (ensure_step_scratch(ctx, A, A)) return -1;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->rows.p, l, A * N3 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xp.p, x_p, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xv.p, x_v, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xa.p, x_a, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, pf, A * 24, hipMemcpyHostToDevice, st));
    if (dmpc_table_from_rows_device(ctx, S, 1, N, ctx->rows.as<double>(), ctx->lT.as<double>(), st)) return -1;
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;
    if (mixed && table_f32(ctx, ctx->lT.as<double>(), ctx->lTf, A * N3, st)) return -1;
    if (launch_step(ctx, S, 1, N, 0, 0, N, ctx->lT.as<double>(), ctx->xp.as<double>(), ctx->xv.as<double>(),
                    ctx->xa.as<double>(), ctx->pf.as<double>(), ctx->pout.as<double>(), ctx->vout.as<double>(),
                    ctx->aout.as<double>(), nullptr, ctx->status.as<int32_t>(), ctx->info.as<int32_t>(), st, nullptr, 0,
                    mixed ? ctx->lTf.as<float>() : nullptr))
        return -1;
    HIPCHK(ctx, hipMemcpyAsync(p_out, ctx->pout.p, A * N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(v_out, ctx->vout.p, A * N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(a_out, ctx->aout.p, A * N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(status, ctx->status.p, A * 4, hipMemcpyDeviceToHost, st));
    if (info) HIPCHK(ctx, hipMemcpyAsync(info, ctx->info.p, A * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

extern "C" int dmpc_solve_one(dmpc_ctx *ctx, int N, int n, const double *l, const double *po, const double *vo,
                              const double *ao, const double *pf, double *p, double *v, double *a, int32_t *status,
                              int32_t *info)
{
    if (!ctx) { g_err = "dmpc_solve_one: ctx is NULL"; return -1; }
    if (N < 1 || n < 0 || n >= N) FAIL(ctx, "dmpc_solve_one: agent index out of range");
    if (!l || !po || !vo || !ao || !pf || !p || !v || !a || !status) FAIL(ctx, "dmpc_solve_one: NULL pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ensure_step_scratch(ctx, (size_t)N, 1)) return -1;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->rows.p, l, (size_t)N * N3 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xp.p, po, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xv.p, vo, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xa.p, ao, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, pf, 24, hipMemcpyHostToDevice, st));
    if (dmpc_table_from_rows_device(ctx, 1, 1, N, ctx->rows.as<double>(), ctx->lT.as<double>(), st)) return -1;
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;
    if (mixed && table_f32(ctx, ctx->lT.as<double>(), ctx->lTf, (size_t)N * N3, st)) return -1;
    if (launch_step(ctx, 1, 1, N, 0, n, 1, ctx->lT.as<double>(), ctx->xp.as<double>(), ctx->xv.as<double>(),
                    ctx->xa.as<double>(), ctx->pf.as<double>(), ctx->pout.as<double>(), ctx->vout.as<double>(),
                    ctx->aout.as<double>(), nullptr, ctx->status.as<int32_t>(), ctx->info.as<int32_t>(), st, nullptr, 0,
                    mixed ? ctx->lTf.as<float>() : nullptr))
        return -1;
    HIPCHK(ctx, hipMemcpyAsync(p, ctx->pout.p, N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(v, ctx->vout.p, N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(a, ctx->aout.p, N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(status, ctx->status.p, 4, hipMemcpyDeviceToHost, st));
    if (info) HIPCHK(ctx, hipMemcpyAsync(info, ctx->info.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

// a5/a6 standalone: scan + collision rows of ONE agent in the reference's row order
// (CheckCollSoftDMPC.m + CollConstrSoftDMPC.m and variants), without pruning.  Host pointers.
extern "C" int dmpc_rows_one(dmpc_ctx *ctx, int N, int n, const double *l, const double *po, const double *vo,
                             int max_rows, double *xi, double *rhs, double *slack_coef, int32_t *kc, int32_t *nrows,
                             int32_t *viol_k, int32_t *status)
{
    if (!ctx) { g_err = "dmpc_rows_one: ctx is NULL"; return -1; }
    if (N < 1 || n < 0 || n >= N || max_rows < 0) FAIL(ctx, "dmpc_rows_one: bad arguments");
    if (!l || !po || !vo || !nrows || !viol_k || !status) FAIL(ctx, "dmpc_rows_one: NULL pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ensure_step_scratch(ctx, (size_t)N, 1)) return -1;
    hipStream_t st = ctx->stream;
    const dmpc_params &p = ctx->prm;
    const bool soft = variant_soft(p.variant);
    double zero3[3] = {0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(ctx->rows.p, l, (size_t)N * N3 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xp.p, po, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xv.p, vo, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->xa.p, zero3, 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, zero3, 24, hipMemcpyHostToDevice, st));
    if (dmpc_table_from_rows_device(ctx, 1, 1, N, ctx->rows.as<double>(), ctx->lT.as<double>(), st)) return -1;
    StepParams P;
    memset(&P, 0, sizeof(P));
    P.variant = p.variant; P.S = 1; P.G = 1; P.C = N; P.g_local = 0; P.c_first = n; P.c_count = 1;
    const long want = (p.variant == DMPC_VAR_HARD ? (long)K : (p.variant == DMPC_VAR_ALL3 ? 3L : 1L)) * (N > 1 ? N - 1 : 1);
    P.nrmax = (int)((want + 1) & ~1L);   // the exact worst case: nothing is pruned or truncated here
    P.ell_order = p.order;
    P.h = p.h; P.rmin = p.rmin; P.e1z = 1.0 / p.c; P.e2z = p.order == 4 ? 1.0 / (p.c * p.c * p.c * p.c) : 1.0 / (p.c * p.c);   // E1 = E^-1, E2 = E^-order
    if (p.variant == DMPC_VAR_SCP) { P.e1z = 1.0; P.e2z = 1.0; }   // solveDMPC: plain Euclidean norm (CheckCollDMPC.m:6, CollConstrDMPC.m:12-13)
    P.alim = p.alim; P.Q1 = p.Q1; P.S1 = p.S1; P.term = p.term;
    for (int d = 0; d < 3; ++d) { P.pmin[d] = p.pmin[d]; P.pmax[d] = p.pmax[d]; }
    P.tables = ctx->d_tables; P.lT = ctx->lT.as<double>();
    for (int i = 0; i < 3; ++i) P.hsum[i] = ctx->hsum[i];
    P.x_p = ctx->xp.as<double>(); P.x_v = ctx->xv.as<double>(); P.x_a = ctx->xa.as<double>(); P.pf = ctx->pf.as<double>();
    P.status = ctx->status.as<int32_t>(); P.no_prune = 1; P.qcap = QMAX; P.qover_bit = ST_CAPACITY;
    const size_t per = (size_t)P.nrmax * (soft ? 7 : 4);
    if (ctx->rowbuf.ensure(per * 8) || ctx->rowkc.ensure((size_t)P.nrmax * 4) || ctx->hdr.ensure(32)) FAIL(ctx, "device allocation failed");
    P.rowbuf = ctx->rowbuf.as<double>(); P.rowkc = ctx->rowkc.as<int>(); P.hdr = ctx->hdr.as<int>();
    P.lds_per_wave = (int)scan_lds_bytes();
    // (order 4 -- the all-neighbour variants only, check_params -- has its own scan kernels: dist = |E1 d|_4, xi = E2 d.^3, prev_dist = dist^3)
    if (p.order == 4) {
        if (soft) hipLaunchKernelGGL((dmpc_scan_kernel<true, double, false, true>), dim3(1), dim3(64), scan_lds_bytes(), st, P);
        else hipLaunchKernelGGL((dmpc_scan_kernel<false, double, false, true>), dim3(1), dim3(64), scan_lds_bytes(), st, P);
    } else if (soft) hipLaunchKernelGGL((dmpc_scan_kernel<true, double, false>), dim3(1), dim3(64), scan_lds_bytes(), st, P);
    else hipLaunchKernelGGL((dmpc_scan_kernel<false, double, false>), dim3(1), dim3(64), scan_lds_bytes(), st, P);
    HIPCHK(ctx, hipGetLastError());
    int hdr[8];
    HIPCHK(ctx, hipMemcpyAsync(hdr, ctx->hdr.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const int nr = hdr[0];
    *nrows = hdr[1]; *viol_k = hdr[2]; *status = hdr[3] & (DMPC_ST_COLL | DMPC_ST_CAPACITY);
    const int cnt = nr < max_rows ? nr : max_rows;
    if (cnt > 0 && xi && rhs && kc) {
        std::vector<double> buf(per);
        std::vector<int> kbuf(P.nrmax);
        HIPCHK(ctx, hipMemcpy(buf.data(), ctx->rowbuf.p, per * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(kbuf.data(), ctx->rowkc.p, (size_t)P.nrmax * 4, hipMemcpyDeviceToHost));
        // reference order: horizon step major, neighbour index minor (the hard variant's rows are emitted
        // neighbour-chunk major on the device): stable sort by constrained step
        std::vector<int> idx(nr);
        for (int i = 0; i < nr; ++i) idx[i] = i;
        if (p.variant == DMPC_VAR_HARD || p.variant == DMPC_VAR_ALL3)
            std::stable_sort(idx.begin(), idx.end(), [&](int a_, int b_) { return kbuf[a_] < kbuf[b_]; });
        for (int i = 0; i < cnt; ++i) {
            const int j = idx[i];
            xi[3 * i] = buf[3 * j]; xi[3 * i + 1] = buf[3 * j + 1]; xi[3 * i + 2] = buf[3 * j + 2];
            rhs[i] = buf[(size_t)3 * P.nrmax + j];
            if (slack_coef) slack_coef[i] = soft ? buf[(size_t)4 * P.nrmax + j] : 0.0;
            kc[i] = kbuf[j] + 1;   // 1-based like the reference's k_ctr
        }
    }
    return 0;
}

extern "C" int dmpc_init_batch(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, double *l_out,
                               double *v_out, double *a_out)
{
    if (!ctx) { g_err = "dmpc_init_batch: ctx is NULL"; return -1; }
    if (S < 1 || N < 1 || !po || !pf || !l_out) FAIL(ctx, "dmpc_init_batch: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t A = (size_t)S * N;
    if (ctx->po.ensure(A * 24) || ctx->pf.ensure(A * 24) || ctx->rows.ensure(A * N3 * 8)) FAIL(ctx, "device allocation failed");
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->po.p, po, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, pf, A * 24, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(init_rows_kernel, dim3((unsigned)((A * N3 + 255) / 256)), dim3(256), 0, st, (int)A, ctx->prm.h,
                       ctx->po.as<double>(), ctx->pf.as<double>(), ctx->rows.as<double>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(l_out, ctx->rows.p, A * N3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (v_out) memset(v_out, 0, A * N3 * 8);   // initDMPC.m:11-12: v = a = zeros(3,k_hor)
    if (a_out) memset(a_out, 0, A * N3 * 8);
    return 0;
}

// the whole `for k = 1:K_T` loop on the device (dmpc_soft_bound.m:115-148, failure_rate.m:99-127)
static int transition_one(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max,
                          double error_tol, double *pk, double *vk, double *ak, int32_t *K_T_used,
                          int32_t *scene_status)
{
    if (!ctx) { g_err = "dmpc_transition: ctx is NULL"; return -1; }
    if (S < 1 || N < 1 || K_T_max < 2 || !po || !pf || !K_T_used || !scene_status || ((pk || vk || ak) && !(pk && vk && ak)))
        FAIL(ctx, "dmpc_transition: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t A = (size_t)S * N;
    if (ensure_step_scratch(ctx, A, A)) return -1;
    const size_t hist = A * (size_t)K_T_max * 24;
    if (ctx->lT2.ensure(A * N3 * 8) || ctx->po.ensure(A * 24) || ctx->hist_p.ensure(hist) || ctx->hist_v.ensure(hist) ||
        ctx->hist_a.ensure(hist) || ctx->flags.ensure((size_t)K_T_max * S * 8) || ctx->scene_done.ensure((size_t)S * 4))
        FAIL(ctx, "device allocation failed");
    hipStream_t st = ctx->stream;
    double *xp = ctx->xp.as<double>(), *xv = ctx->xv.as<double>(), *xa = ctx->xa.as<double>();
    HIPCHK(ctx, hipMemcpyAsync(ctx->po.p, po, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, pf, A * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->flags.p, 0, (size_t)K_T_max * S * 8, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->scene_done.p, 0, (size_t)S * 4, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_p.p, 0, hist, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_v.p, 0, hist, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->hist_a.p, 0, hist, st));
    // k = 1: initDMPC (dmpc_soft_bound.m:117-121): state = (po, 0, 0), table = straight lines
    HIPCHK(ctx, hipMemcpyAsync(xp, ctx->po.p, A * 24, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(xv, 0, A * 24, st));
    HIPCHK(ctx, hipMemsetAsync(xa, 0, A * 24, st));
    hipLaunchKernelGGL(init_rows_kernel, dim3((unsigned)((A * N3 + 255) / 256)), dim3(256), 0, st, (int)A, ctx->prm.h,
                       ctx->po.as<double>(), ctx->pf.as<double>(), ctx->rows.as<double>());
    if (dmpc_table_from_rows_device(ctx, S, 1, N, ctx->rows.as<double>(), ctx->lT.as<double>(), st)) return -1;
    const unsigned rb = (unsigned)((A * 3 + 255) / 256);
    hipLaunchKernelGGL(record_kernel, dim3(rb), dim3(256), 0, st, S, N, K_T_max, 0, xp, xv, xa, ctx->hist_p.as<double>(),
                       ctx->hist_v.as<double>(), ctx->hist_a.as<double>());
    // ReachedGoal is also evaluated on the initDMPC column (failure_rate.m:125 runs after k = 1 as well)
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->status.p, DMPC_ST_SOLVED, A, st));
    hipLaunchKernelGGL(scene_reduce_kernel, dim3((unsigned)S), dim3(256), 0, st, N, error_tol, xp, ctx->pf.as<double>(),
                       (const int *)ctx->status.as<int32_t>(), ctx->flags.as<int>(), ctx->scene_done.as<int>());
    double *cur = ctx->lT.as<double>(), *nxt = ctx->lT2.as<double>();
    ctx->post_acc_S = 0;   // the scene accumulators of the fused post-step start from zero in every transition
    const bool mixed = (ctx->precision & DMPC_PREC_MIXED) != 0;   // the scan of every step reads an fp32 copy of the current table
    // The host looks at the per-step verdicts every `chunk` MPC steps -- one window BEHIND the steps it enqueues: the verdicts of window c
    // are copied to pinned host memory behind an event, the steps of window c+1 are enqueued, and only then the host waits for the event
    // of window c.  The device never idles while the host reads flags (a stream synchronisation per window cost 40-45 us of idle GPU per
    // 8 steps: 8 % of a single-scene transition); the steps enqueued past the end of a trial are skipped on the device (scene_done), so
    // what a scene records and reports does not change.
    const size_t flag_ints = (size_t)K_T_max * S * 2;
    if (ctx->flags_host_cap < flag_ints) {
        if (ctx->flags_host) (void)hipHostFree(ctx->flags_host);
        ctx->flags_host = nullptr; ctx->flags_host_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->flags_host, flag_ints * 4, hipHostMallocDefault));
        ctx->flags_host_cap = flag_ints;
    }
    int32_t *flags = ctx->flags_host;
    for (int u = 0; u < 2; ++u)
        if (!ctx->flag_ev[u]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->flag_ev[u], hipEventDisableTiming));
    std::vector<int> done(S, 0);
    for (int s = 0; s < S; ++s) { K_T_used[s] = K_T_max; scene_status[s] = DMPC_ST_SOLVED; }
    int ndone = 0;
    const int chunk = 8;
    int pend_k0 = -1, pend_k1 = -1, pend_ev = 0, nwin = 0;   // the window whose verdicts are in flight to the host
    auto digest = [&]() -> int {   // wait for the pending window and fold its verdicts into done / K_T_used / scene_status
        if (pend_k0 < 0) return 0;
        HIPCHK(ctx, hipEventSynchronize(ctx->flag_ev[pend_ev]));
        for (int kk = pend_k0; kk <= pend_k1; ++kk)
            for (int s = 0; s < S; ++s) {
                if (done[s]) continue;
                const int32_t reached = flags[((size_t)kk * S + s) * 2], stbits = flags[((size_t)kk * S + s) * 2 + 1];
                if (stbits & ~DMPC_ST_SOLVED) {   // some agent failed: the reference aborts the trial
                    done[s] = 1; ndone++; K_T_used[s] = kk + 1; scene_status[s] = stbits;
                } else if (reached) {             // ReachedGoal.m (failure_rate.m:125)
                    done[s] = 1; ndone++; K_T_used[s] = kk + 1; scene_status[s] = DMPC_ST_SOLVED | DMPC_ST_REACHED;
                }
            }
        pend_k0 = -1;
        return 0;
    };
    for (int k = 1; k < K_T_max && ndone < S; ++k) {
        if (mixed && table_f32(ctx, cur, ctx->lTf, A * N3, st)) return -1;
        const PostStep post{K_T_max, k, error_tol, xp, xv, xa, ctx->hist_p.as<double>(), ctx->hist_v.as<double>(), ctx->hist_a.as<double>(),
                            ctx->flags.as<int>() + (size_t)k * S * 2, ctx->scene_done.as<int>()};
        if (launch_step(ctx, S, 1, N, 0, 0, N, cur, xp, xv, xa, ctx->pf.as<double>(), ctx->pout.as<double>(),
                        ctx->vout.as<double>(), ctx->aout.as<double>(), nxt, ctx->status.as<int32_t>(), nullptr, st,
                        ctx->scene_done.as<int>(), 0, mixed ? ctx->lTf.as<float>() : nullptr, &post))
            return -1;
        // state advance + history column + scene verdict in one launch (unless the solve kernel did them: tiny launches)
        if (!ctx->post_fused)
        hipLaunchKernelGGL(post_step_kernel, dim3((unsigned)S), dim3(N >= 256 ? 256 : 128), 0, st, N, K_T_max, k, error_tol,
                           (const double *)ctx->pout.as<double>(), (const double *)ctx->vout.as<double>(), (const double *)ctx->aout.as<double>(),
                           (const int *)ctx->status.as<int32_t>(), xp, xv, xa, (const double *)ctx->pf.as<double>(), ctx->hist_p.as<double>(),
                           ctx->hist_v.as<double>(), ctx->hist_a.as<double>(), ctx->flags.as<int>() + (size_t)k * S * 2, ctx->scene_done.as<int>(),
                           (const int *)ctx->scene_done.as<int>());
        HIPCHK(ctx, hipGetLastError());
        std::swap(cur, nxt);   // l = new_l (dmpc_soft_bound.m:146)
        if (k % chunk == 0 || k == K_T_max - 1) {
            if (digest()) return -1;                                         // the window before this one (its steps ran while this one was enqueued)
            const int k0 = k <= chunk ? 0 : ((k - 1) / chunk) * chunk + 1;   // (the first window includes the initDMPC column)
            HIPCHK(ctx, hipMemcpyAsync(&flags[(size_t)k0 * S * 2], ctx->flags.as<int>() + (size_t)k0 * S * 2,
                                       (size_t)(k - k0 + 1) * S * 8, hipMemcpyDeviceToHost, st));
            pend_ev = nwin++ & 1;
            HIPCHK(ctx, hipEventRecord(ctx->flag_ev[pend_ev], st));
            pend_k0 = k0; pend_k1 = k;
        }
    }
    if (digest()) return -1;
    if (pk) {   // the histories stay resident for dmpc_postcheck either way; the download is optional
        HIPCHK(ctx, hipMemcpyAsync(pk, ctx->hist_p.p, hist, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(vk, ctx->hist_v.p, hist, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(ak, ctx->hist_a.p, hist, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->hist_S = S; ctx->hist_N = N; ctx->hist_KT = K_T_max;
    return 0;
}

// development options of a context handed to a context it creates for itself (further parts of a split batch, the second group)
static void copy_debug_options(dmpc_ctx *dst, const dmpc_ctx *src)
{
    static const char *names[] = {"no_fuse", "no_persist", "force_persist", "no_cull", "order_slices", "cull_min", "no_lpt", "crash_min", "crash_any", "no_fast_exit",
                                  "pivot_explore", "iter_cap", "tier1_qcap", "static_queue", "queue_chunk", "no_split_t", "ext_cap", "nbr_grid", "f32_dep_exp", "grid_min", "no_level_check", "order_hint", "lds_pad_kb", "reduced_solver", "rsolve_cap", "no_level_skip", "prep_fuse"};
    int dmpc_ctx::*fields[] = {&dmpc_ctx::no_fuse, &dmpc_ctx::no_persist, &dmpc_ctx::force_persist, &dmpc_ctx::no_cull, &dmpc_ctx::order_slices, &dmpc_ctx::cull_min,
                               &dmpc_ctx::no_lpt, &dmpc_ctx::crash_min, &dmpc_ctx::crash_any, &dmpc_ctx::no_fast_exit, &dmpc_ctx::pivot_explore, &dmpc_ctx::iter_cap,
                               &dmpc_ctx::tier1_env, &dmpc_ctx::static_queue, &dmpc_ctx::queue_chunk, &dmpc_ctx::no_split_t, &dmpc_ctx::ext_cap, &dmpc_ctx::nbr_grid, &dmpc_ctx::f32_dep_exp, &dmpc_ctx::grid_min, &dmpc_ctx::no_level_check, &dmpc_ctx::order_hint, &dmpc_ctx::lds_pad_kb, &dmpc_ctx::reduced_solver, &dmpc_ctx::rsolve_cap, &dmpc_ctx::no_level_skip, &dmpc_ctx::prep_fuse};
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i) (void)dmpc_debug_option(dst, names[i], src->*(fields[i]));
}

// Batched transitions are bound, MPC step by MPC step, by the slowest agent of the whole batch while most of the GPU
// idles.  Scenes are independent, so a large batch is run as two halves on two contexts (= two HIP streams, two host
// threads): the tail of one half overlaps the bulk of the other (512 transitions of 100 agents: 103 -> 60 ms).
extern "C" int dmpc_transition(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max,
                               double error_tol, double *pk, double *vk, double *ak, int32_t *K_T_used,
                               int32_t *scene_status)
{
    if (!ctx) { g_err = "dmpc_transition: ctx is NULL"; return -1; }
    ctx->split_at.clear();
    if (ctx->grp) {   // every visible GPU: the agents of each scene sharded over them (dmpc_multigpu.hip)
        if (S < 1 || N < 1 || K_T_max < 2 || !po || !pf || !K_T_used || !scene_status || ((pk || vk || ak) && !(pk && vk && ak))) FAIL(ctx, "dmpc_transition: bad arguments");
        // Batches of 64 or more scenes run as TWO groups side by side (a second set of rank contexts, threads and streams on the same
        // GPUs): scenes are independent, so while one half's ranks exchange their predictions (peer copies, barrier, events) the
        // other half's solve kernels keep the GPUs busy -- the per-step exchange is off the critical path.
        const int gparts = ctx->no_split ? 1 : (ctx->split_parts > 0 ? (ctx->split_parts > 2 ? 2 : ctx->split_parts) : (S >= 64 ? 2 : 1));
        // fewer agents than twice the GPUs (the reference's small swarms on an 8-GPU node): the first GPU alone, as dmpc_step_batch does --
        // sharding N = 4 agents over 8 GPUs is impossible and over 2-3 of them nothing but barriers and peer copies
        if (N < 2 * ctx->grp->G) return transition_one(ctx, S, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status);
        if (gparts < 2) return group_transition(ctx, S, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status);
        if (ctx->children.empty()) {
            const int keep = g_emulate_devices.load();
            if (ctx->grp_emulated) g_emulate_devices.store(ctx->grp->G);
            dmpc_ctx *other = dmpc_create(&ctx->prm, DMPC_DEVICE_ALL, ctx->precision);
            g_emulate_devices.store(keep);
            if (!other || !other->grp || other->grp->G != ctx->grp->G) { if (other) dmpc_destroy(other); FAIL(ctx, "dmpc_transition: second group: " + g_err); }
            copy_debug_options(other, ctx);   // (and, through dmpc_debug_option, to its rank contexts)
            other->no_split = 1;
            ctx->children.push_back(other);
        }
        dmpc_ctx *other = ctx->children[0];
        if (std::memcmp(&other->prm, &ctx->prm, sizeof(dmpc_params)) != 0 && dmpc_set_params(other, &ctx->prm)) FAIL(ctx, "dmpc_transition: second group: " + other->err);
        const int S0 = S / 2;
        const size_t a0 = (size_t)S0 * N, h0 = a0 * (size_t)K_T_max * 3;
        int rc1 = 0;
        std::thread th([&]() {
            rc1 = group_transition(other, S - S0, N, po + a0 * 3, pf + a0 * 3, K_T_max, error_tol, pk ? pk + h0 : nullptr, vk ? vk + h0 : nullptr,
                                   ak ? ak + h0 : nullptr, K_T_used + S0, scene_status + S0);
        });
        const int rc0 = group_transition(ctx, S0, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status);
        th.join();
        (void)hipSetDevice(ctx->device);
        if (rc1) FAIL(ctx, other->err);
        if (rc0) return -1;
        ctx->split_at = {0, S0, S};
        ctx->hist_S = S;
        return 0;
    }
    // parts: every part runs its own MPC loop on its own stream; launches of fewer than ~2000 agents are one-agent workgroups, which the
    // hardware interleaves across streams freely (persistent launches hold a CU's whole LDS), so many small parts overlap best
    int parts = ctx->split_parts > 0 ? ctx->split_parts : (S >= 128 ? 4 : (S >= 32 ? 2 : 1));
    if (parts > S) parts = S;
    if (parts < 2 || ctx->no_split || N < 1 || K_T_max < 2 || !po || !pf || !K_T_used || !scene_status)
        return transition_one(ctx, S, N, po, pf, K_T_max, error_tol, pk, vk, ak, K_T_used, scene_status);
    while ((int)ctx->children.size() < parts - 1) {
        dmpc_ctx *ch = dmpc_create(&ctx->prm, ctx->device, ctx->precision);
        if (!ch) FAIL(ctx, "dmpc_transition: further context: " + g_err);
        copy_debug_options(ch, ctx);
        ch->no_split = 1;
        ctx->children.push_back(ch);
    }
    for (int i = 0; i < parts - 1; ++i)
        if (std::memcmp(&ctx->children[(size_t)i]->prm, &ctx->prm, sizeof(dmpc_params)) != 0 && dmpc_set_params(ctx->children[(size_t)i], &ctx->prm))
            FAIL(ctx, "dmpc_transition: further context: " + ctx->children[(size_t)i]->err);
    std::vector<int> at((size_t)parts + 1);
    for (int i = 0; i <= parts; ++i) at[(size_t)i] = (int)((long)S * i / parts);
    std::vector<int> rc((size_t)parts, 0);
    auto run = [&](int i) {
        dmpc_ctx *c = i ? ctx->children[(size_t)i - 1] : ctx;
        const int s0 = at[(size_t)i], sn = at[(size_t)i + 1] - s0;
        const size_t a0 = (size_t)s0 * N, h0 = a0 * (size_t)K_T_max * 3;
        rc[(size_t)i] = transition_one(c, sn, N, po + a0 * 3, pf + a0 * 3, K_T_max, error_tol, pk ? pk + h0 : nullptr, vk ? vk + h0 : nullptr,
                                       ak ? ak + h0 : nullptr, K_T_used + s0, scene_status + s0);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < parts; ++i) th.emplace_back(run, i);
    run(0);
    for (auto &t : th) t.join();
    for (int i = 1; i < parts; ++i)
        if (rc[(size_t)i]) FAIL(ctx, ctx->children[(size_t)i - 1]->err);
    if (rc[0]) return -1;
    ctx->split_at = at;   // the resident histories are split over the contexts (dmpc_postcheck knows)
    ctx->hist_S = S;
    return 0;
}

// post-checks of S finished transitions (failure_rate.m:136-195): rescale, 100 Hz not-a-knot spline, pairwise
// ellipsoidal collision check, path length, trajectory time.  pk == NULL: use the histories dmpc_transition left
// resident on the device (no PCIe round trip).
static int postcheck_one(dmpc_ctx *ctx, int S, int N, int KT_alloc, const int32_t *K_T_used, const int32_t *scene_mask,
                         const double *pk, const double *vk, const double *ak, const double *pf, double vmax, double amax, double Ts,
                         double *r_factor, double *h_scaled, int32_t *n_samples, double *min_dist,
                         int32_t *violation, double *totdist, double *traj_time, double *p_interp, int ns_alloc)
{
    if (!ctx) { g_err = "dmpc_postcheck: ctx is NULL"; return -1; }
    if (S < 1 || N < 1 || KT_alloc < 2 || !K_T_used || !pf || !(vmax > 0) || !(amax > 0) || !(Ts > 0))
        FAIL(ctx, "dmpc_postcheck: bad arguments");
    std::vector<int32_t> kt(S);
    for (int s = 0; s < S; ++s) {
        const bool on = !scene_mask || scene_mask[s];
        if (on && (K_T_used[s] < 2 || K_T_used[s] > KT_alloc)) FAIL(ctx, "dmpc_postcheck: K_T_used out of range");
        kt[s] = on ? K_T_used[s] : 0;   // masked scenes (aborted trials, failure_rate.m:136) are skipped by every kernel
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t A = (size_t)S * N, hist = A * (size_t)KT_alloc * 24;
    if (ctx->pc_p.ensure(hist) || ctx->pc_v.ensure(hist) || ctx->pc_a.ensure(hist) || ctx->pc_M.ensure(hist) ||
        ctx->pc_w.ensure(hist) || ctx->pc_scene.ensure((size_t)S * 64) || ctx->pc_agent.ensure(A * 16) || ctx->pf.ensure(A * 24))
        FAIL(ctx, "device allocation failed");
    if (pk) {
        if (!vk || !ak) FAIL(ctx, "dmpc_postcheck: vk/ak missing");
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_p.p, pk, hist, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_v.p, vk, hist, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_a.p, ak, hist, hipMemcpyHostToDevice, st));
    } else {
        if (ctx->hist_S != S || ctx->hist_N != N || ctx->hist_KT != KT_alloc)
            FAIL(ctx, "dmpc_postcheck: no resident histories of this shape (run dmpc_transition first or pass pk/vk/ak)");
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_p.p, ctx->hist_p.p, hist, hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_v.p, ctx->hist_v.p, hist, hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pc_a.p, ctx->hist_a.p, hist, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf.p, pf, A * 24, hipMemcpyHostToDevice, st));
    // per-scene scalars: [0] kt_used(int) [1] rf [2] hs [3] ns(int) [4] mind2(u64) [5] totdist [6] traj_time
    char *sc = ctx->pc_scene.as<char>();
    int *d_kt = (int *)sc;
    double *d_rf = (double *)(sc + (size_t)S * 8), *d_hs = (double *)(sc + (size_t)S * 16);
    int *d_ns = (int *)(sc + (size_t)S * 24);
    unsigned long long *d_min = (unsigned long long *)(sc + (size_t)S * 32);
    double *d_tot = (double *)(sc + (size_t)S * 40), *d_tt = (double *)(sc + (size_t)S * 48);
    HIPCHK(ctx, hipMemcpyAsync(d_kt, kt.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(d_min, 0x7f, (size_t)S * 8, st));   // 0x7f7f... = a huge finite double
    double *dp = ctx->pc_p.as<double>(), *dv = ctx->pc_v.as<double>(), *da = ctx->pc_a.as<double>();
    hipLaunchKernelGGL(pc::rfactor_kernel, dim3((unsigned)S), dim3(256), 0, st, N, KT_alloc, (const int *)d_kt, (const double *)dv,
                       (const double *)da, vmax, amax, d_rf);
    std::vector<double> rf(S), hs(S), md(S), tot(S), tt(S);
    std::vector<int32_t> ns(S);
    HIPCHK(ctx, hipMemcpyAsync(rf.data(), d_rf, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    int ns_max = 0;
    for (int s = 0; s < S; ++s) {
        if (!kt[s]) { rf[s] = hs[s] = NAN; ns[s] = 0; continue; }
        if (!(std::isfinite(rf[s]) && rf[s] > 0))   // MATLAB: h_scaled = 0, tk = 0:0:T is empty and spline() errors
            FAIL(ctx, "dmpc_postcheck: degenerate r_factor (all-zero or non-finite histories)");
        hs[s] = ctx->prm.h / std::sqrt(rf[s]);                                   // failure_rate.m:146
        const double T = (kt[s] - 1) * hs[s];                              // :149
        ns[s] = (std::isfinite(T) && T / Ts < 5e7) ? (int)std::floor(T / Ts + 1e-10) + 1 : -1;   // :152
        if (ns[s] < 1) FAIL(ctx, "dmpc_postcheck: degenerate r_factor (all-zero or non-finite histories)");
        ns_max = std::max(ns_max, (int)ns[s]);
    }
    HIPCHK(ctx, hipMemcpyAsync(d_hs, hs.data(), (size_t)S * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_ns, ns.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
    const unsigned b3 = (unsigned)((A * 3 + 255) / 256);
    hipLaunchKernelGGL(pc::rescale_kernel, dim3(b3), dim3(256), 0, st, S, N, KT_alloc, (const int *)d_kt, (const double *)d_rf,
                       (const double *)d_hs, dp, dv, da);
    hipLaunchKernelGGL(pc::spline_kernel, dim3(b3), dim3(256), 0, st, S, N, KT_alloc, (const int *)d_kt, (const double *)d_hs,
                       (const double *)dp, ctx->pc_M.as<double>(), ctx->pc_w.as<double>());
    double *d_interp = nullptr;
    if (p_interp) {
        if (ns_alloc < 1) FAIL(ctx, "dmpc_postcheck: ns_alloc must be positive with p_interp");
        if (ctx->pc_interp.ensure(A * (size_t)ns_alloc * 24)) FAIL(ctx, "device allocation failed");
        d_interp = ctx->pc_interp.as<double>();
        HIPCHK(ctx, hipMemsetAsync(d_interp, 0, A * (size_t)ns_alloc * 24, st));
    }
    // grid search of the large scenes: `grid_pass(scene_on)` runs every sample batch; scene_on = null: cell-grid search of all
    // scenes, else brute force over the same batch positions for the scenes flagged in it
    bool use_grid = N > PC_BRUTE_MAX && ns_max > 0;
    pc::Grid g{};
    double edge = 2.0 * ctx->prm.rmin;
    int SB = 1, ncell = 1;
    if (use_grid) {
        const dmpc_params &pr = ctx->prm;
        for (;;) {   // cells of `edge` (z: edge * c) over the workspace + 2 cells of margin, at most 32768 of them
            g.nx = (int)std::ceil((pr.pmax[0] - pr.pmin[0]) / edge) + 2;
            g.ny = (int)std::ceil((pr.pmax[1] - pr.pmin[1]) / edge) + 2;
            g.nz = (int)std::ceil((pr.pmax[2] - pr.pmin[2]) / (edge * pr.c)) + 2;
            if ((double)g.nx * g.ny * g.nz <= 32768.0) break;
            edge *= 1.25;
        }
        g.x0 = pr.pmin[0] - edge; g.y0 = pr.pmin[1] - edge; g.z0 = pr.pmin[2] - edge * pr.c;
        g.inv_e = 1.0 / edge; g.inv_ez = 1.0 / (edge * pr.c);
        ncell = g.nx * g.ny * g.nz;
        const double per_sample = (double)S * ((double)N * 36.0 + (double)ncell * 8.0 + 4.0);
        SB = (int)std::floor(256.0 * 1048576.0 / per_sample);
        SB = SB < 1 ? 1 : (SB > 256 ? 256 : SB);
        if (SB > ns_max) SB = ns_max;
        const size_t sbn = (size_t)S * SB;
        if (ctx->pc_pts.ensure(sbn * N * 24) || ctx->pc_cell.ensure(sbn * N * 4) || ctx->pc_sorted.ensure(sbn * N * 4) ||
            ctx->pc_fill.ensure(sbn * ncell * 4) || ctx->pc_start.ensure(sbn * ((size_t)ncell + 1) * 4) || ctx->pc_on.ensure((size_t)S * 4))
            FAIL(ctx, "device allocation failed (post-check cell grid)");
        HIPCHK(ctx, hipMemsetAsync(ctx->pc_fill.p, 0, sbn * ncell * 4, st));
    }
    auto grid_pass = [&](const int *scene_on) -> int {
        const size_t sbn = (size_t)S * SB, tot = sbn * N;
        for (int smp0 = 0; smp0 < ns_max; smp0 += SB) {
            hipLaunchKernelGGL(pc::grid_eval_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, S, N, KT_alloc, (const int *)d_kt,
                               (const double *)d_hs, (const int *)d_ns, Ts, smp0, SB, (const double *)dp, (const double *)ctx->pc_M.as<double>(), g,
                               ctx->pc_pts.as<double>(), ctx->pc_cell.as<int>(), ctx->pc_fill.as<int>(), scene_on ? (double *)nullptr : d_interp, ns_alloc);
            if (!scene_on) {
                hipLaunchKernelGGL(pc::grid_scan_kernel, dim3((unsigned)sbn), dim3(1024), 0, st, ncell, ctx->pc_fill.as<int>(), ctx->pc_start.as<int>());
                hipLaunchKernelGGL(pc::grid_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tot, N, ncell,
                                   (const int *)ctx->pc_cell.as<int>(), (const int *)ctx->pc_start.as<int>(), ctx->pc_fill.as<int>(), ctx->pc_sorted.as<int>());
                HIPCHK(ctx, hipMemsetAsync(ctx->pc_fill.p, 0, sbn * ncell * 4, st));
                hipLaunchKernelGGL(pc::grid_pairs_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)SB, (unsigned)S), dim3(256), 0, st, N, SB, g,
                                   1.0 / ctx->prm.c, (const double *)ctx->pc_pts.as<double>(), (const int *)ctx->pc_cell.as<int>(),
                                   (const int *)ctx->pc_start.as<int>(), (const int *)ctx->pc_sorted.as<int>(), d_min);
            } else {
                HIPCHK(ctx, hipMemsetAsync(ctx->pc_fill.p, 0, sbn * ncell * 4, st));   // (the evaluation counted again)
                hipLaunchKernelGGL(pc::pairs_brute_pts_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)SB, (unsigned)S), dim3(256), 0, st, N, SB,
                                   1.0 / ctx->prm.c, (const double *)ctx->pc_pts.as<double>(), (const int *)ctx->pc_cell.as<int>(), scene_on, d_min);
            }
            HIPCHK(ctx, hipGetLastError());
        }
        return 0;
    };
    if (use_grid) { if (grid_pass(nullptr)) return -1; }
    else if (ns_max > 0)
    hipLaunchKernelGGL(pc::pairdist_kernel, dim3((unsigned)((ns_max + PC_SAMPLES_PER_BLOCK - 1) / PC_SAMPLES_PER_BLOCK), (unsigned)S),
                       dim3(256), (size_t)N * 24, st, N, KT_alloc, (const int *)d_kt, (const double *)d_hs, (const int *)d_ns, Ts,
                       1.0 / ctx->prm.c, (const double *)dp, (const double *)ctx->pc_M.as<double>(), d_min, d_interp, ns_alloc);
    double *d_dist = ctx->pc_agent.as<double>();
    int *d_tidx = (int *)(ctx->pc_agent.as<char>() + A * 8);
    hipLaunchKernelGGL(pc::path_kernel, dim3((unsigned)((A + 63) / 64)), dim3(64), 0, st, S, N, KT_alloc, (const int *)d_kt,
                       (const double *)d_hs, (const int *)d_ns, Ts, (const double *)dp, (const double *)ctx->pc_M.as<double>(),
                       (const double *)ctx->pf.as<double>(), d_dist, d_tidx);
    hipLaunchKernelGGL(pc::finish_kernel, dim3((unsigned)S), dim3(256), 0, st, N, (const double *)d_dist, (const int *)d_tidx, Ts,
                       d_tot, d_tt);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(md.data(), d_min, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    if (use_grid) {
        // scenes in which the grid found no pair closer than its cell edge: nothing is closer than that (no violation); the exact
        // minimum, which the interface reports, takes the brute-force search over the same sample batches
        HIPCHK(ctx, hipStreamSynchronize(st));
        std::vector<int> on(S, 0);
        int n_on = 0;
        for (int s = 0; s < S; ++s)
            if (kt[s] && N > 1 && !(md[s] <= edge * edge)) { on[s] = 1; n_on++; }
        ctx->pc_fallback_scenes = n_on;
        if (n_on) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->pc_on.p, on.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
            if (grid_pass(ctx->pc_on.as<int>())) return -1;
            HIPCHK(ctx, hipMemcpyAsync(md.data(), d_min, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(tot.data(), d_tot, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(tt.data(), d_tt, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    if (p_interp) HIPCHK(ctx, hipMemcpyAsync(p_interp, d_interp, A * (size_t)ns_alloc * 24, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int s = 0; s < S; ++s) {
        if (!kt[s]) {
            if (r_factor) r_factor[s] = NAN;
            if (h_scaled) h_scaled[s] = NAN;
            if (n_samples) n_samples[s] = 0;
            if (min_dist) min_dist[s] = NAN;
            if (violation) violation[s] = 0;
            if (totdist) totdist[s] = NAN;
            if (traj_time) traj_time[s] = NAN;
            continue;
        }
        const double d = (N > 1) ? std::sqrt(md[s]) : INFINITY;
        if (r_factor) r_factor[s] = rf[s];
        if (h_scaled) h_scaled[s] = hs[s];
        if (n_samples) n_samples[s] = ns[s];
        if (min_dist) min_dist[s] = d;
        if (violation) violation[s] = d < ctx->prm.rmin - 0.05;                  // failure_rate.m:175
        if (totdist) totdist[s] = tot[s];
        if (traj_time) traj_time[s] = tt[s];
    }
    return 0;
}

extern "C" int dmpc_postcheck(dmpc_ctx *ctx, int S, int N, int KT_alloc, const int32_t *K_T_used, const int32_t *scene_mask,
                              const double *pk, const double *vk, const double *ak, const double *pf, double vmax, double amax, double Ts,
                              double *r_factor, double *h_scaled, int32_t *n_samples, double *min_dist,
                              int32_t *violation, double *totdist, double *traj_time, double *p_interp, int ns_alloc)
{
    if (!ctx) { g_err = "dmpc_postcheck: ctx is NULL"; return -1; }
    const int parts = (int)ctx->split_at.size() - 1;
    if (pk || parts < 2 || ctx->split_at.back() != S || ctx->hist_S != S || (int)ctx->children.size() < parts - 1 || !K_T_used || !pf)
        return postcheck_one(ctx, S, N, KT_alloc, K_T_used, scene_mask, pk, vk, ak, pf, vmax, amax, Ts, r_factor, h_scaled, n_samples,
                             min_dist, violation, totdist, traj_time, p_interp, ns_alloc);
    // histories left resident by a split dmpc_transition: each part is checked where it lives, concurrently
    auto off = [&](auto *ptr, size_t o) { return ptr ? ptr + o : ptr; };
    std::vector<int> rc((size_t)parts, 0);
    auto run = [&](int i) {
        dmpc_ctx *c = i ? ctx->children[(size_t)i - 1] : ctx;
        const int s0 = ctx->split_at[(size_t)i], sn = ctx->split_at[(size_t)i + 1] - s0;
        const size_t a0 = (size_t)s0 * N;
        const int keep = c->hist_S;
        c->hist_S = sn;
        rc[(size_t)i] = postcheck_one(c, sn, N, KT_alloc, K_T_used + s0, off(scene_mask, (size_t)s0), nullptr, nullptr, nullptr, pf + a0 * 3, vmax, amax, Ts,
                                      off(r_factor, (size_t)s0), off(h_scaled, (size_t)s0), off(n_samples, (size_t)s0), off(min_dist, (size_t)s0),
                                      off(violation, (size_t)s0), off(totdist, (size_t)s0), off(traj_time, (size_t)s0),
                                      off(p_interp, a0 * (size_t)ns_alloc * 3), ns_alloc);
        c->hist_S = keep;
    };
    std::vector<std::thread> th;
    for (int i = 1; i < parts; ++i) th.emplace_back(run, i);
    run(0);
    for (auto &t : th) t.join();
    for (int i = 1; i < parts; ++i)
        if (rc[(size_t)i]) FAIL(ctx, ctx->children[(size_t)i - 1]->err);
    return rc[0];
}


// ---------------------------------------------------------------------------------------------
// dense collision-row builders (dec-iSCP/CollConstr.m, dmpc/matlab/CollConstr*DMPC.m, cup-SCP/AddCollConstr.m)
// ---------------------------------------------------------------------------------------------
static size_t strided_extent(int rows, int cols, int64_t rs, int64_t cs)
{
    return (size_t)((rows - 1) * rs + (cols - 1) * cs + 1);
}

extern "C" int dmpc_coll_rows_device(dmpc_ctx *ctx, int K, int n_sel, const int32_t *d_sel, const double *d_l, int k_cmp, int k_blk,
                                     const double *p, const double *a0, double rmin, double c, const double *d_A, int64_t a_rs,
                                     int64_t a_cs, int ncols, double *d_Ain, int64_t o_rs, int64_t o_cs, double *d_bin,
                                     double *d_dist, void *stream)
{
    if (!ctx) { g_err = "dmpc_coll_rows_device: ctx is NULL"; return -1; }
    if (K < 1 || n_sel < 0 || !p || !a0 || k_cmp < 0 || k_cmp >= K || k_blk < 0 || ncols < 1 || !(c > 0) || !d_A || !d_Ain || !d_bin)
        FAIL(ctx, "dmpc_coll_rows_device: bad arguments");
    if (n_sel == 0) return 0;
    if (!d_sel || !d_l) FAIL(ctx, "dmpc_coll_rows_device: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t tot = (size_t)n_sel * ncols;
    hipLaunchKernelGGL(rb::coll_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_sel, (const int *)d_sel, K, d_l, k_cmp, k_blk, p[0], p[1], p[2], a0[0], a0[1], a0[2], rmin, 1.0 / c, d_A,
                       (long)a_rs, (long)a_cs, ncols, d_Ain, (long)o_rs, (long)o_cs, d_bin, d_dist, ctx->prm.order == 4 ? 4 : 2);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C" int dmpc_coll_rows(dmpc_ctx *ctx, int K, int N_obs, int n_sel, const int32_t *sel, const double *l, int k_cmp, int k_blk,
                              const double *p, const double *a0, double rmin, double c, const double *A, int a_rows, int ncols,
                              int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs, double *bin, double *dist)
{
    if (!ctx) { g_err = "dmpc_coll_rows: ctx is NULL"; return -1; }
    if (K < 1 || N_obs < 0 || n_sel < 0 || n_sel > N_obs || a_rows < 3 || ncols < 1 || a_rs < 1 || a_cs < 1 || o_rs < 1 || o_cs < 1 ||
        !A || !Ain || !bin || !p || !a0)
        FAIL(ctx, "dmpc_coll_rows: bad arguments");
    if (3 * k_blk + 2 >= a_rows || k_blk < 0) FAIL(ctx, "dmpc_coll_rows: constraint block outside A");
    if (n_sel == 0) return 0;
    if (!sel || !l) FAIL(ctx, "dmpc_coll_rows: bad arguments");
    for (int i = 0; i < n_sel; ++i)
        if (sel[i] < 0 || sel[i] >= N_obs) FAIL(ctx, "dmpc_coll_rows: obstacle index out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t a_ext = strided_extent(a_rows, ncols, a_rs, a_cs), o_ext = strided_extent(n_sel, ncols, o_rs, o_cs);
    if (ctx->rb_A.ensure(a_ext * 8) || ctx->rb_l.ensure((size_t)N_obs * K * 24) || ctx->rb_sel.ensure((size_t)n_sel * 4) ||
        ctx->rb_out.ensure(o_ext * 8) || ctx->rb_bin.ensure((size_t)n_sel * 16))
        FAIL(ctx, "device allocation failed");
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_A.p, A, a_ext * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_l.p, l, (size_t)N_obs * K * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_sel.p, sel, (size_t)n_sel * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->rb_out.p, 0, o_ext * 8, st));
    double *d_bin = ctx->rb_bin.as<double>(), *d_dist = d_bin + n_sel;
    if (dmpc_coll_rows_device(ctx, K, n_sel, ctx->rb_sel.as<int32_t>(), ctx->rb_l.as<double>(), k_cmp, k_blk, p, a0, rmin, c,
                              ctx->rb_A.as<double>(), a_rs, a_cs, ncols, ctx->rb_out.as<double>(), o_rs, o_cs, d_bin, d_dist, st))
        return -1;
    HIPCHK(ctx, hipMemcpyAsync(Ain, ctx->rb_out.p, o_ext * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(bin, d_bin, (size_t)n_sel * 8, hipMemcpyDeviceToHost, st));
    if (dist) HIPCHK(ctx, hipMemcpyAsync(dist, d_dist, (size_t)n_sel * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

extern "C" int dmpc_rows_dense(dmpc_ctx *ctx, int nr, const double *xi, const int32_t *kc, const double *A, int a_rows, int ncols,
                               int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs)
{
    if (!ctx) { g_err = "dmpc_rows_dense: ctx is NULL"; return -1; }
    if (nr < 0 || a_rows < 3 || ncols < 1 || a_rs < 1 || a_cs < 1 || o_rs < 1 || o_cs < 1 || !A || !Ain) FAIL(ctx, "dmpc_rows_dense: bad arguments");
    if (nr == 0) return 0;
    if (!xi || !kc) FAIL(ctx, "dmpc_rows_dense: bad arguments");
    for (int r = 0; r < nr; ++r)
        if (kc[r] < 1 || 3 * kc[r] > a_rows) FAIL(ctx, "dmpc_rows_dense: constraint block outside A");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t a_ext = strided_extent(a_rows, ncols, a_rs, a_cs), o_ext = strided_extent(nr, ncols, o_rs, o_cs);
    if (ctx->rb_A.ensure(a_ext * 8) || ctx->rb_l.ensure((size_t)nr * 24) || ctx->rb_sel.ensure((size_t)nr * 4) || ctx->rb_out.ensure(o_ext * 8))
        FAIL(ctx, "device allocation failed");
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_A.p, A, a_ext * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_l.p, xi, (size_t)nr * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_sel.p, kc, (size_t)nr * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->rb_out.p, 0, o_ext * 8, st));
    const size_t tot = (size_t)nr * ncols;
    hipLaunchKernelGGL(rb::xi_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, nr, (const double *)ctx->rb_l.as<double>(),
                       (const int *)ctx->rb_sel.as<int>(), (const double *)ctx->rb_A.as<double>(), (long)a_rs, (long)a_cs, ncols,
                       ctx->rb_out.as<double>(), (long)o_rs, (long)o_cs);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(Ain, ctx->rb_out.p, o_ext * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

extern "C" int dmpc_add_coll_constr_device(dmpc_ctx *ctx, int K, int N, const double *d_p, const double *d_po, double rmin, double c,
                                           const double *d_A, int64_t a_rs, int64_t a_cs, int ncols, double *d_Ain, int64_t o_rs,
                                           int64_t o_cs, double *d_bin, void *stream)
{
    if (!ctx) { g_err = "dmpc_add_coll_constr_device: ctx is NULL"; return -1; }
    if (K < 1 || N < 2 || ncols < 1 || !(c > 0) || !d_p || !d_po || !d_A || !d_Ain || !d_bin)
        FAIL(ctx, "dmpc_add_coll_constr_device: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nrows = (size_t)K * N * (N - 1) / 2, tot = nrows * (size_t)ncols;
    if ((tot + 255) / 256 > 0x7fffffffull) FAIL(ctx, "dmpc_add_coll_constr_device: problem too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    const size_t pch = ((size_t)N * (N - 1) / 2 + RB_PCH - 1) / RB_PCH;
    if (o_cs == 1 && K <= 65535 && pch <= 65535) {   // row-major output: (column tile, k, pair chunk) blocks
        hipLaunchKernelGGL(rb::add_coll_rows_rm_kernel, dim3((unsigned)((ncols + 255) / 256), (unsigned)K, (unsigned)pch), dim3(256), 0, st,
                           N, K, d_p, d_po, rmin, 1.0 / c, d_A, (long)a_rs, (long)a_cs, ncols, d_Ain, (long)o_rs, d_bin, ctx->prm.order == 4 ? 4 : 2);
    } else if (o_rs == 1 && (ncols + RB_CCH - 1) / RB_CCH <= 65535) {   // column-major output (MATLAB): row-per-thread
        hipLaunchKernelGGL(rb::add_coll_rows_cm_kernel, dim3((unsigned)((nrows + 255) / 256), (unsigned)((ncols + RB_CCH - 1) / RB_CCH)),
                           dim3(256), 0, st, N, K, d_p, d_po, rmin, 1.0 / c, d_A, (long)a_rs, (long)a_cs, ncols, d_Ain, (long)o_cs,
                           d_bin, nrows, ctx->prm.order == 4 ? 4 : 2);
    } else {                                // arbitrary strides: one element per thread
        hipLaunchKernelGGL(rb::add_coll_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, K, d_p, d_po, rmin,
                           1.0 / c, d_A, (long)a_rs, (long)a_cs, ncols, d_Ain, (long)o_rs, (long)o_cs, d_bin, nrows, ctx->prm.order == 4 ? 4 : 2);
    }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C" int dmpc_add_coll_constr(dmpc_ctx *ctx, int K, int N, const double *p, const double *po, double rmin, double c,
                                    const double *A, int ncols, int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs,
                                    double *bin)
{
    if (!ctx) { g_err = "dmpc_add_coll_constr: ctx is NULL"; return -1; }
    if (K < 1 || N < 2 || ncols < 1 || a_rs < 1 || a_cs < 1 || o_rs < 1 || o_cs < 1 || !p || !po || !A || !Ain || !bin)
        FAIL(ctx, "dmpc_add_coll_constr: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int a_rows = 3 * K * N;
    const size_t nrows = (size_t)K * N * (N - 1) / 2;
    const size_t a_ext = strided_extent(a_rows, ncols, a_rs, a_cs);
    const size_t o_ext = (size_t)((nrows - 1) * o_rs + (size_t)(ncols - 1) * o_cs + 1);
    if (ctx->rb_A.ensure(a_ext * 8) || ctx->rb_l.ensure((size_t)N * K * 24) || ctx->rb_po.ensure((size_t)N * 24) ||
        ctx->rb_out.ensure(o_ext * 8) || ctx->rb_bin.ensure(nrows * 8))
        FAIL(ctx, "device allocation failed");
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_A.p, A, a_ext * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_l.p, p, (size_t)N * K * 24, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_po.p, po, (size_t)N * 24, hipMemcpyHostToDevice, st));
    if (dmpc_add_coll_constr_device(ctx, K, N, ctx->rb_l.as<double>(), ctx->rb_po.as<double>(), rmin, c, ctx->rb_A.as<double>(), a_rs,
                                    a_cs, ncols, ctx->rb_out.as<double>(), o_rs, o_cs, ctx->rb_bin.as<double>(), st))
        return -1;
    HIPCHK(ctx, hipMemcpyAsync(Ain, ctx->rb_out.p, o_ext * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(bin, ctx->rb_bin.p, nrows * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

#include "dmpc_fileio.hip"
#include "dmpc_multigpu.hip"


// ---------------------------------------------------------------------------------------------
// start / goal generators (randomTest.m, randomExchange.m)
// ---------------------------------------------------------------------------------------------
static int random_sets(dmpc_ctx *ctx, const char *who, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                       uint64_t seed, int exchange, double *d_out, hipStream_t st)
{
    if (S < 1 || N < 1 || !pmin || !pmax || !(rmin >= 0) || !(c > 0)) FAIL(ctx, std::string(who) + ": bad arguments");
    for (int d = 0; d < 3; ++d) if (!(pmax[d] > pmin[d])) FAIL(ctx, std::string(who) + ": empty box");
    const size_t lds = (size_t)N * 24 + (exchange ? (size_t)N * 8 : 0);
    if (lds > 150 * 1024) FAIL(ctx, std::string(who) + ": at most ~4800 agents per scene");
    HIPCHK(ctx, hipFuncSetAttribute((const void *)gen::random_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int waves = exchange ? S : 2 * S;
    hipLaunchKernelGGL(gen::random_points_kernel, dim3((unsigned)waves), dim3(64), lds, st, S, N, 2, exchange, pmin[0], pmin[1], pmin[2],
                       pmax[0], pmax[1], pmax[2], rmin, 1.0 / c, seed, d_out);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

static int random_host(dmpc_ctx *ctx, const char *who, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                       uint64_t seed, int exchange, double *po, double *pf)
{
    if (!ctx) { g_err = std::string(who) + ": ctx is NULL"; return -1; }
    if (!po || !pf || S < 1 || N < 1) FAIL(ctx, std::string(who) + ": bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t one = (size_t)S * N * 24;
    if (ctx->gen_out.ensure(2 * one)) FAIL(ctx, "device allocation failed");
    if (random_sets(ctx, who, S, N, pmin, pmax, rmin, c, seed, exchange, ctx->gen_out.as<double>(), ctx->stream)) return -1;
    HIPCHK(ctx, hipMemcpyAsync(po, ctx->gen_out.p, one, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pf, ctx->gen_out.as<char>() + one, one, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int dmpc_random_test(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                                uint64_t seed, double *po, double *pf)
{
    return random_host(ctx, "dmpc_random_test", S, N, pmin, pmax, rmin, c, seed, 0, po, pf);
}

extern "C" int dmpc_random_exchange(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, uint64_t seed,
                                    double *po, double *pf)
{
    return random_host(ctx, "dmpc_random_exchange", S, N, pmin, pmax, rmin, 1.0, seed, 1, po, pf);
}

extern "C" int dmpc_random_sets_device(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                                       uint64_t seed, int exchange, double *d_po_pf, void *stream)
{
    if (!ctx) { g_err = "dmpc_random_sets_device: ctx is NULL"; return -1; }
    if (!d_po_pf) FAIL(ctx, "dmpc_random_sets_device: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return random_sets(ctx, "dmpc_random_sets_device", S, N, pmin, pmax, rmin, exchange ? 1.0 : c, seed, exchange ? 1 : 0, d_po_pf,
                       (hipStream_t)stream);
}


// ---------------------------------------------------------------------------------------------
// standalone forms of the small helpers of the path (propStatedmpc.m, dec-iSCP/propState.m, is_inbounds.m,
// ReachedGoal.m): inside the solvers they are fused into the step kernels; callers that invoke them on their own
// (the reference's scripts do) get the same arithmetic on the device.  Host pointers, synchronous.
// ---------------------------------------------------------------------------------------------
namespace hp {
// p = A_p a + A_initp x0 + tile(off_p), v = A_v a + tile(off_v); one thread per output row
__global__ void prop_state_kernel(int n_rows, int n_cols, const double *__restrict__ A_p, const double *__restrict__ A_v,
                                  const double *__restrict__ A_initp, const double *__restrict__ x0, const double *__restrict__ off,
                                  const double *__restrict__ a, double *__restrict__ p, double *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    double sp = 0.0, sv = 0.0;
    for (int j = 0; j < n_cols; ++j) {
        sp += A_p[(size_t)i * n_cols + j] * a[j];
        sv += A_v[(size_t)i * n_cols + j] * a[j];
    }
    if (A_initp) {
        double s0 = 0.0;
        for (int u = 0; u < 6; ++u) s0 += A_initp[(size_t)i * 6 + u] * x0[u];
        sp += s0;
    }
    p[i] = sp + off[i % 3];
    v[i] = sv + off[3 + i % 3];
}
// is_inbounds.m:2-5: every coordinate of every point strictly inside [pmin - 5 cm, pmax + 5 cm]
__global__ void inbounds_kernel(int npts, const double *__restrict__ p, const double *__restrict__ lim, int *__restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * npts) return;
    const int d = i % 3;
    const double tol = 50e-3;
    if (!(p[i] < lim[3 + d] + tol) || !(p[i] > lim[d] - tol)) atomicOr(bad, 1);
}
// ReachedGoal.m:2-10: max_i |p_i - pf_i| < error_tol
__global__ void reached_kernel(int N, const double *__restrict__ p, const double *__restrict__ pf, double tol, int *__restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double dx = p[3 * i] - pf[3 * i], dy = p[3 * i + 1] - pf[3 * i + 1], dz = p[3 * i + 2] - pf[3 * i + 2];
    if (!(sqrt(dx * dx + dy * dy + dz * dz) < tol)) atomicOr(bad, 1);
}
// maxDeviation.m:3-9: per step the distance between the two trajectories (non-negative doubles order like their bit patterns: integer maximum)
__global__ void max_dev_kernel(int nsteps, const double *__restrict__ p, const double *__restrict__ q, unsigned long long *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsteps) return;
    const double dx = p[3 * i] - q[3 * i], dy = p[3 * i + 1] - q[3 * i + 1], dz = p[3 * i + 2] - q[3 * i + 2];
    atomicMax(out, (unsigned long long)__double_as_longlong(sqrt(dx * dx + dy * dy + dz * dz)));
}
}   // namespace hp

extern "C" int dmpc_max_deviation(dmpc_ctx *ctx, int K_cols, const double *p, const double *prev_p, double *tol_out)
{
    if (!ctx) { g_err = "dmpc_max_deviation: ctx is NULL"; return -1; }
    if (K_cols < 1 || !p || !prev_p || !tol_out) FAIL(ctx, "dmpc_max_deviation: bad arguments");
    // `K = length(p)/3; for k = 1:K` on the 3 x K_cols matrix (maxDeviation.m:3-5): length() = max(3, K_cols), the loop visits floor(that / 3) columns
    const int nsteps = (K_cols > 3 ? K_cols : 3) / 3;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (ctx->hp_in.ensure((size_t)6 * nsteps * 8) || ctx->hp_out.ensure(16)) FAIL(ctx, "device allocation failed");
    double *dp = ctx->hp_in.as<double>(), *dq = dp + 3 * nsteps;
    unsigned long long *out = ctx->hp_out.as<unsigned long long>();
    HIPCHK(ctx, hipMemcpyAsync(dp, p, (size_t)3 * nsteps * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dq, prev_p, (size_t)3 * nsteps * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(out, 0, 8, st));
    hipLaunchKernelGGL(hp::max_dev_kernel, dim3((unsigned)((nsteps + 63) / 64)), dim3(64), 0, st, nsteps, (const double *)dp, (const double *)dq, out);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long bits = 0;
    HIPCHK(ctx, hipMemcpyAsync(&bits, out, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    std::memcpy(tol_out, &bits, 8);
    return 0;
}

extern "C" int dmpc_prop_state(dmpc_ctx *ctx, int n_rows, int n_cols, const double *A_p, const double *A_v, const double *A_initp,
                               const double *po, const double *vo, const double *off_p, const double *off_v, const double *a,
                               double *p, double *v)
{
    if (!ctx) { g_err = "dmpc_prop_state: ctx is NULL"; return -1; }
    if (n_rows < 1 || n_cols < 1 || !A_p || !A_v || !a || !p || !v || (A_initp && (!po || !vo))) FAIL(ctx, "dmpc_prop_state: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nA = (size_t)n_rows * n_cols;
    // layout of the staging buffer: A_p | A_v | A_initp | x0(6) | off(6) | a
    const size_t tot = 2 * nA + (size_t)n_rows * 6 + 12 + n_cols;
    if (ctx->hp_in.ensure(tot * 8) || ctx->hp_out.ensure((size_t)n_rows * 16 + 16)) FAIL(ctx, "device allocation failed");
    double *d = ctx->hp_in.as<double>();
    double *dAp = d, *dAv = d + nA, *dA0 = dAv + nA, *dx0 = dA0 + (size_t)n_rows * 6, *doff = dx0 + 6, *da = doff + 6;
    double x0[6] = {0, 0, 0, 0, 0, 0}, off[6] = {0, 0, 0, 0, 0, 0};
    for (int u = 0; u < 3; ++u) {
        if (po) x0[u] = po[u];
        if (vo) x0[3 + u] = vo[u];
        if (off_p) off[u] = off_p[u];
        if (off_v) off[3 + u] = off_v[u];
    }
    HIPCHK(ctx, hipMemcpyAsync(dAp, A_p, nA * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dAv, A_v, nA * 8, hipMemcpyHostToDevice, st));
    if (A_initp) HIPCHK(ctx, hipMemcpyAsync(dA0, A_initp, (size_t)n_rows * 48, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dx0, x0, 48, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(doff, off, 48, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(da, a, (size_t)n_cols * 8, hipMemcpyHostToDevice, st));
    double *dp = ctx->hp_out.as<double>(), *dv = dp + n_rows;
    hipLaunchKernelGGL(hp::prop_state_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, st, n_rows, n_cols, (const double *)dAp,
                       (const double *)dAv, A_initp ? (const double *)dA0 : (const double *)nullptr, (const double *)dx0,
                       (const double *)doff, (const double *)da, dp, dv);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(p, dp, (size_t)n_rows * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(v, dv, (size_t)n_rows * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

static int flag_kernel_common(dmpc_ctx *ctx, const char *who, int count3, const double *p, int extra_n, const double *extra,
                              int which, double tol, int32_t *out)
{
    if (!ctx) { g_err = std::string(who) + ": ctx is NULL"; return -1; }
    if (count3 < 1 || !p || !extra || !out) FAIL(ctx, std::string(who) + ": bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (ctx->hp_in.ensure(((size_t)count3 + extra_n) * 8) || ctx->hp_out.ensure(16)) FAIL(ctx, "device allocation failed");
    double *dp = ctx->hp_in.as<double>(), *dx = dp + count3;
    int *bad = ctx->hp_out.as<int>();
    HIPCHK(ctx, hipMemcpyAsync(dp, p, (size_t)count3 * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(dx, extra, (size_t)extra_n * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(bad, 0, 4, st));
    if (which == 0)
        hipLaunchKernelGGL(hp::inbounds_kernel, dim3((unsigned)((count3 + 255) / 256)), dim3(256), 0, st, count3 / 3, (const double *)dp,
                           (const double *)dx, bad);
    else
        hipLaunchKernelGGL(hp::reached_kernel, dim3((unsigned)((count3 / 3 + 255) / 256)), dim3(256), 0, st, count3 / 3, (const double *)dp,
                           (const double *)dx, tol, bad);
    HIPCHK(ctx, hipGetLastError());
    int32_t b = 0;
    HIPCHK(ctx, hipMemcpyAsync(&b, bad, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    *out = b ? 0 : 1;
    return 0;
}

extern "C" int dmpc_is_inbounds(dmpc_ctx *ctx, int npts, const double *p, const double *pmin, const double *pmax, int32_t *inbounds)
{
    if (!pmin || !pmax) { g_err = "dmpc_is_inbounds: bad arguments"; return -1; }
    const double lim[6] = {pmin[0], pmin[1], pmin[2], pmax[0], pmax[1], pmax[2]};
    return flag_kernel_common(ctx, "dmpc_is_inbounds", 3 * npts, p, 6, lim, 0, 0.0, inbounds);
}

extern "C" int dmpc_reached_goal(dmpc_ctx *ctx, int N, const double *p, const double *pf, double error_tol, int32_t *reached)
{
    return flag_kernel_common(ctx, "dmpc_reached_goal", 3 * N, p, 3 * N, pf, 1, error_tol, reached);
}
