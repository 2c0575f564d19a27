/*
 * dmpc_hip.h -- C ABI of libdmpc_hip.so: the MI355X (gfx950) implementation of the DMPC
 * per-agent horizon-QP hot path of carlosluis/multiagent_planning.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository root).  Plain C, plain pointers and sizes: this is what a MEX gateway,
 * a cgo/ctypes stub or the C++ DMPC class would bind.
 *
 * Conventions
 *   - all real data is IEEE fp64 (MATLAB double / C++ double), K = k_hor = 15
 *   - "table" l: the previous MPC step's predicted position horizons of all agents.
 *     Host layout == MATLAB l(3,K,N) column-major == row-major [N][3K] with the stacked
 *     [x1 y1 z1 x2 ...] order (dmpc/matlab/dmpc_soft_bound.m:131,146).
 *   - S independent scenes ("trials", test/failure_rate.m:65) may be batched: arrays are
 *     [S][N][...]; agents only see neighbours of their own scene.  S = 1 is the reference case.
 *   - return value: 0 = ok, <0 = API/runtime error (text via dmpc_last_error); numerical
 *     outcomes are reported per agent in status[] (DMPC_ST_* bits) and info[] (DMPC_I_*).
 *   - entry points are synchronous unless noted; one HIP stream per context; a context may be
 *     used from one host thread at a time and has ONE step in flight at a time (the *_device entry
 *     points share per-context scratch: do not issue them on two streams of one context
 *     concurrently); distinct contexts are independent (dmpc/cpp/cluster_test.cpp:40 runs up to
 *     10 solver threads).
 */
#ifndef DMPC_HIP_H
#define DMPC_HIP_H

#include <stdint.h>

/* the library is built with hidden visibility: the entry points declared here (and the development interface of
 * dmpc_hip_dev.h) are all it exports */
#define DMPC_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

/* solver variants == reference functions under dmpc/matlab/ */
enum {
    DMPC_VAR_BOUND = 0,    /* solveSoftDMPCbound.m:1        (primary; test/failure_rate.m:110)  */
    DMPC_VAR_BOUND2 = 1,   /* solveSoftDMPCbound2.m:1       (test/comp_kctr.m:248)              */
    DMPC_VAR_ALL3 = 2,     /* solveSoftDMPCall.m:1                                              */
    DMPC_VAR_HARD = 3,     /* solveHardDMPC.m:1             (test/comp_hardsoft2.m)             */
    DMPC_VAR_ONDEMAND = 4, /* solveHardDMPCOnDemand.m:1                                         */
    DMPC_VAR_ELLIP = 5,    /* solveEllipDMPC.m:1                                                */
    DMPC_VAR_SOFTALL = 6,  /* solveSoftDMPC.m:1             (test/success_test_softdmpc.m:90)   */
    DMPC_VAR_REPAIR = 7,   /* solveSoftDMPCrepair.m:1       (test/comp_repair.m:93)             */
    /* the C++ flavour, DMPC::solveQPv2 (dmpc/cpp/dmpc.cpp:803-1287) with _k_factor = 0 / -1 (main.cpp:37-38):
     * bound / bound2 with the near-neighbour radius rmin*(1+(float)k/k_hor) (:418), slack bounds [-0.01f, 0] doubled on
     * retry together with term (<= 20 retries, :1079-1088), no early return on a first-step collision (DMPC_ST_COLL is
     * reported NEXT TO a solution, :419-424) and no in-bounds test.  All members of the C++ class are floats
     * (dmpc.h:191-205): pass h, rmin, c, alim as (double)(float) values and term = -1e6, Q1 = 1000, S1 = 100 (:846,942-945)
     * to reproduce its arithmetic. */
    DMPC_VAR_CPP = 8,
    DMPC_VAR_CPP2 = 9,
    /* the FIRST C++ version, DMPC::solveQP (dmpc/cpp/dmpc.cpp:554-801; callers solveDMPC :1367, cluster_solve :1776): at the first
     * horizon step k (0-based) where check_collisions (:378-396) finds a neighbour inside rmin, rows for ALL N-1 neighbours
     * (build_collconstraint :450-498) on step k-1, each with its own slack: [A I; 0 I] x <= [b; 0] (:629-633), linear slack cost -1e6
     * (:715), quadratic slack weight 1 (:719); cost cases by `violation` (:640-661); ONE QuadProgDense solve, no retry, no in-bounds
     * test, no first-step collision test.  A violation at k = 0 makes the reference index row -3 of A0 (undefined behaviour): reported
     * as DMPC_ST_COLL without a solve.  Float members as for DMPC_VAR_CPP. */
    DMPC_VAR_CPP1 = 10,
    /* solveSoftDMPC_c.m:1-96 (test/comp_confidence.m:184): solveSoftDMPC with the slack penalties of :60-63 -- linear -1e4 (K/k)^2, quadratic
     * 1e6 (K/k)^2, k the first violating horizon step (1-based) -- and no outbound output.  order 2 or 4. */
    DMPC_VAR_SOFTALL_C = 11,
    /* solveDMPC.m:1-74, the legacy SCP loop (dmpc/matlab/dmpc.m:79, test/success_test_dmpc.m:80, test/comp_heuristics.m:76): up to k_hor
     * passes; every pass re-linearises HARD spherical rows (CheckCollDMPC.m:6-8: plain Euclidean norm, no E1; CollConstrDMPC.m:8-30) for ALL
     * N-1 neighbours at every horizon step of `addConstr` about the previous pass's prediction, adds at most ONE new step per pass (the first
     * violating step not yet in the set, :28-33), solves the QP (cost case by `isempty(Ain_total)` only, :38-48) and stops when
     * maxDeviation(p, prev_p) <= dmpc_params.tol (:17,69; maxDeviation.m:3 looks at the first length(p)/3 = 5 steps of the 3 x 15 matrix).
     * The whole loop of an agent runs inside ONE kernel launch.  info: VIOLK = smallest step of addConstr, NROWS = rows of the last pass,
     * TRIES = passes made, CASE = 0 / 2 of the last pass; an infeasible pass ends the loop with DMPC_ST_INFEAS (`success = 0`, :58-63).
     * c and order are not used (sphere); no in-bounds test, no first-step collision test. */
    DMPC_VAR_SCP = 12
};

/* per-agent status bits (the reference's feasible/success, outbound, coll flags) */
enum {
    DMPC_ST_SOLVED = 1,    /* p,v,a valid                                                        */
    DMPC_ST_OUTBOUND = 2,  /* is_inbounds.m:2-5 failed for the first predicted position          */
    DMPC_ST_COLL = 4,      /* already collided at horizon step 1 (solveSoftDMPCbound.m:25-31)    */
    DMPC_ST_INFEAS = 8,    /* QP infeasible after the retry ladder (solveSoftDMPCbound.m:102-155) */
    DMPC_ST_CAPACITY = 16, /* internal capacity exceeded (rows / active set): result NOT valid   */
    DMPC_ST_ITERCAP = 32,  /* iteration cap hit: result NOT valid                                */
    DMPC_ST_REACHED = 256  /* scene_status of dmpc_transition only: all agents reached their goals */
};

/* info[] : 8 int32 per agent */
enum {
    DMPC_I_VIOLK = 0,   /* 1-based first violating horizon step handled (0 = none)              */
    DMPC_I_NROWS = 1,   /* number of collision rows built (Nv)                                  */
    DMPC_I_TRIES = 2,   /* QP attempts (retry ladder)                                           */
    DMPC_I_CASE = 3,    /* cost case 0 far / 1 near / 2 collision (solveSoftDMPCbound.m:43-58)  */
    DMPC_I_ITERS = 4,   /* active-set iterations (all tries)                                    */
    DMPC_I_NSLACK = 5,  /* slack variables < 0 at the solution                                  */
    DMPC_I_NACTIVE = 6, /* active constraints at the solution                                   */
    DMPC_I_MAXQ = 7,    /* peak working-set size                                                */
    DMPC_INFO_LEN = 8
};

/* mirrors the reference's constants block (dmpc/matlab/dmpc_soft_bound.m:7-78;
 * dmpc/cpp/dmpc.h:50-63 `struct Params`) */
typedef struct {
    int32_t K;         /* horizon length k_hor; must be 15                                   */
    int32_t variant;   /* DMPC_VAR_*                                                         */
    int32_t order;     /* ellipsoid order: 2; or 4 (super-ellipsoid of test/comp_test_ellipconstr.m:158-187: dist = |E1 d|_4, E2 = E^-4) with
                        * the all-neighbour variants DMPC_VAR_SOFTALL / _SOFTALL_C / _ELLIP / _REPAIR / _CPP1; refused otherwise */
    int32_t max_tries; /* <=0: reference default (30)                                        */
    double h;          /* time step                                                          */
    double rmin;       /* collision radius                                                   */
    double c;          /* E = diag(1,1,c)                                                    */
    double alim;       /* |a| limit                                                          */
    double Q1, S1;     /* collision-case weights                                             */
    double term;       /* linear slack penalty (negative)                                    */
    double pmin[3], pmax[3];
    /* Weights of the two collision-free cost cases.  0 = the constants hard-coded at the reference's HEAD
     * (solveSoftDMPCbound.m:44-52, dmpc.cpp:922-938): far from the goal Q = 1000, within 1 m Q = 10000, S = 10 in both.
     * Earlier revisions of dmpc/cpp used other values: the recorded dmpc/cpp_results/trajectories (200-agents).txt is
     * reproduced to its 6 printed digits with Qfar = 100, Qnear = 1000 (tests/test_oracle_golden.py). */
    double Qfar, Qnear, Sfree;
    double tol;        /* DMPC_VAR_SCP only: `tol` of solveDMPC(..., Delta, tol, Q1, S1) (solveDMPC.m:1): the SCP loop ends when the largest
                        * position change of a pass falls to tol or below.  Ignored by every other variant.  (new in ABI revision 6) */
} dmpc_params;

typedef struct dmpc_ctx dmpc_ctx;

/* arithmetic of a context */
enum {
    DMPC_PREC_F64 = 0,   /* everything fp64 (the reference is MATLAB double)                                          */
    DMPC_PREC_MIXED = 1, /* the prediction table is kept in fp32 and the scan + collision rows (a5/a6) are computed in
                          * fp32; the QP itself (cost, factor, multipliers, propagation) and all inputs / outputs stay
                          * fp64.  Every entry point takes such a context: the host-pointer ones, the device-pointer steps (they
                          * make the fp32 copy of the caller's fp64 table themselves) and the sharded transitions, where the table
                          * the ranks exchange per step is the fp32 one (half the payload).  BASELINE configs[4].               */
    DMPC_PREC_F32FACTOR = 2, /* the QP below fp64 (BASELINE configs[4], "fp32 vs fp64 tolerance sweep"): the inverse factor of the
                          * active-set solver -- its largest object, what every iteration multiplies with -- is STORED in fp32; cost,
                          * multipliers, residuals and the refinement of the result stay fp64 (lambda += T T' rho until the active-set
                          * residual is at 1e-13 or stops contracting).  Scan and rows fp64.  Not bit-compatible with DMPC_PREC_F64:
                          * the sweep (tests/test_gpu_precision.py) reports status agreement and l_inf per variant. */
    DMPC_PREC_LOW = 3    /* DMPC_PREC_MIXED | DMPC_PREC_F32FACTOR: fp32 table / scan / rows AND fp32 factor                      */
    /* DMPC_VAR_ALL3 keeps the fp64 factor whatever the context's precision says: its three nearly parallel rows per neighbour are the
     * one case the sweep found the fp32 factor unfit for (1 % of its agent-steps on another retry-ladder level). */
};

/* Create a solver context.  Replaces the constants/precompute preamble of dmpc/matlab/dmpc_soft_bound.m:80-108 and the DMPC ctor
 * dmpc/cpp/dmpc.cpp:19-75.  precision: DMPC_PREC_*.
 *   device >= 0           one HIP device.
 *   DMPC_DEVICE_ALL (-100) EVERY visible GPU, from this one process: the agents of each scene are sharded over the GPUs in the
 *                         reference's contiguous thread clusters (DMPC::solveParallelDMPCv2, dmpc/cpp/dmpc.cpp:1600-1625: N/G each,
 *                         the first N mod G one more), one internal host thread + stream + sub-context per GPU, and the join of
 *                         every MPC step (`prev_obs = obs`, :1671-1681; `l = new_l`, dmpc_soft_bound.m:146) is a set of direct
 *                         peer copies over xGMI, each rank writing its chunk of new predictions into every GPU's next table
 *                         (one address space: no collective library involved; the one-process-per-GPU form below uses RCCL).
 *                         The host-pointer entry points a MATLAB / C++ caller uses -- dmpc_transition, dmpc_step_batch,
 *                         dmpc_postcheck on the resident histories -- work unchanged on such a context; the per-agent and helper
 *                         entry points run on its first GPU; the device-pointer entry points (one device's memory) refuse it.
 *                         With one visible GPU this is a plain context.  Results do not depend on the number of GPUs, bit for bit.
 *   DMPC_DEVICE_CURRENT (-1) the calling thread's current HIP device (one process per GPU: "this rank's GPU").
 *   Any other negative value is refused.
 * Returns NULL on failure (no device, bad parameters); dmpc_last_error(NULL) has the text. */
#define DMPC_DEVICE_CURRENT (-1)   /* (the value "current device" has had since the first version of this header) */
#define DMPC_DEVICE_ALL (-100)
/* ABI revision of this header.  The special device values changed once (the round-3 header had DMPC_DEVICE_ALL = -1, DMPC_DEVICE_CURRENT = -2;
 * since revision 4 they are the two values above and any other negative device is refused), so a binding compiled against another header
 * should compare dmpc_abi_version() with the DMPC_ABI_VERSION it was built with before its first dmpc_create. */
#define DMPC_ABI_VERSION 7
DMPC_API int dmpc_abi_version(void);
DMPC_API dmpc_ctx *dmpc_create(const dmpc_params *prm, int device, int precision);
/* number of GPUs the context drives (1 unless created with DMPC_DEVICE_ALL on a multi-GPU node) */
DMPC_API int dmpc_group_size(const dmpc_ctx *ctx);
DMPC_API void dmpc_destroy(dmpc_ctx *ctx);
DMPC_API const char *dmpc_last_error(const dmpc_ctx *ctx);

/* Change parameters (variant, weights, bounds ...) of an existing context. */
DMPC_API int dmpc_set_params(dmpc_ctx *ctx, const dmpc_params *prm);

/* a1-a3: getPosMat.m:1 (Lambda = A = A_p), A_v / A_initp loop dmpc_soft_bound.m:92-108,
 * getDeltaMat.m:1.  Host computation, row-major; any pointer may be NULL.
 * Lambda, Av, Delta: 3K x 3K; A0: 3K x 6. */
DMPC_API int dmpc_model_matrices(const dmpc_params *prm, double *Lambda, double *Av, double *A0, double *Delta);

/* getPosVelMat.m:1 (dec-iSCP/cup-SCP helper kept for signature parity): Aaug (12 x 3K). */
DMPC_API int dmpc_posvel_matrix(double h, int K, double *Aaug);

/* a4: initDMPC.m:1 for S*N agents (MPC step k = 1).  Host pointers.
 * po, pf: [S][N][3]; l_out, v_out, a_out: [S][N][3K]. */
DMPC_API int dmpc_init_batch(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, double *l_out,
                    double *v_out, double *a_out);

/* a11: ONE MPC step for all agents of S scenes == the body of `for n = 1:N`
 * (dmpc_soft_bound.m:116-135 / cluster_solvev2 dmpc/cpp/dmpc.cpp:1792-1841) with the solver
 * selected by prm.variant.  Host pointers; copies in, launches, copies out, synchronises.
 * l: [S][N][3K]; x_p,x_v,x_a (= pk,vk,ak(:,k-1,n)), pf: [S][N][3];
 * p_out,v_out,a_out: [S][N][3K] (rows of agents without DMPC_ST_SOLVED are zero);
 * status: [S][N]; info: [S][N][8] (may be NULL). */
DMPC_API int dmpc_step_batch(dmpc_ctx *ctx, int S, int N, const double *l, const double *x_p, const double *x_v,
                    const double *x_a, const double *pf, double *p_out, double *v_out, double *a_out,
                    int32_t *status, int32_t *info);

/* a7/a8: one agent (0-based n) of one scene: the per-call entry the signature-preserving MATLAB
 * wrappers use ([p,v,a,feasible,outbound,coll] = solveSoftDMPCbound(...), solveSoftDMPCbound.m:1).
 * Host pointers.  l: [N][3K]; po,vo,ao,pf: [3]; p,v,a: [3K]; info: [8] or NULL. */
DMPC_API int dmpc_solve_one(dmpc_ctx *ctx, int N, int n, const double *l, const double *po, const double *vo,
                   const double *ao, const double *pf, double *p, double *v, double *a, int32_t *status,
                   int32_t *info);

/* a5/a6 standalone: the scan and the collision rows of ONE agent (0-based n) exactly as the selected
 * solver variant builds them -- [violation,min_dist,viol_constr] = CheckCollSoftDMPC(...) (CheckCollSoftDMPC.m:1)
 * followed by [Ain,bin,prev_dist] = CollConstrSoftDMPC(...) (CollConstrSoftDMPC.m:1; ...2 / Hard / HardOnDemand /
 * Ellip variants) -- in structured form, reference row order, no pruning.  Host pointers.
 *   row i:  -xi_i' * Lambda(3kc_i-2:3kc_i, :) * a  [+ slack_coef_i * eps_i]  <=  rhs_i
 *   xi: [max_rows][3] = E2*(p - p_j); kc: 1-based constrained horizon step; rhs = bin; slack_coef = prev_dist
 *   (1 for solveSoftDMPC, 0 for the hard variants); nrows = number of rows built (may exceed max_rows);
 *   viol_k = 1-based first violating horizon step (0 none); status = DMPC_ST_COLL or 0. */
DMPC_API int dmpc_rows_one(dmpc_ctx *ctx, int N, int n, const double *l, const double *po, const double *vo, int max_rows,
                  double *xi, double *rhs, double *slack_coef, int32_t *kc, int32_t *nrows, int32_t *viol_k,
                  int32_t *status);

/* Device-resident form of a11 for callers that keep state in HBM (bench, multi-GPU driver).
 * The table is in the chunked transposed layout lT[G][S][3K][C]: G chunks (= ranks) of C agents
 * each, N = G*C agents per scene; this call solves the C agents of chunk g_local of every scene
 * against the whole table (the per-step all-gather concatenates the ranks' lT_next chunks).
 * All pointers are DEVICE pointers; `stream` is the caller's hipStream_t and is used as given (NULL = HIP's
 * default stream, which is also PyTorch's default current stream), so the launches are ordered with the caller's
 * other work on that stream (e.g. the RCCL all-gather of lT_next); asynchronous: returns after enqueueing.
 *   x_p,x_v,x_a,pf : [S][C][3]          p_out,v_out,a_out : [S][C][3K]
 *   lT_next        : [S][3K][C] (this chunk, may be NULL)   status [S][C], info [S][C][8] */
DMPC_API int dmpc_step_device(dmpc_ctx *ctx, int S, int G, int C, int g_local, const double *lT, const double *x_p,
                     const double *x_v, const double *x_a, const double *pf, double *p_out, double *v_out,
                     double *a_out, double *lT_next, int32_t *status, int32_t *info, void *stream);

/* layout helpers (device pointers, asynchronous on `stream`):
 * rows [S][N][3K] -> lT [G][S][3K][C] with N = G*C, and first columns x_next = out(:,1). */
DMPC_API int dmpc_table_from_rows_device(dmpc_ctx *ctx, int S, int G, int C, const double *rows, double *lT, void *stream);

/* Fused state advance on device: x_p,x_v,x_a <- first horizon column of p_out,v_out,a_out for
 * SOLVED agents (dmpc_soft_bound.m:132-134).  [S][C] agents. */
DMPC_API int dmpc_advance_device(dmpc_ctx *ctx, int count, const double *p_out, const double *v_out, const double *a_out,
                        const int32_t *status, double *x_p, double *x_v, double *x_a, void *stream);

/* Whole transition on one device: the `for k = 1:K_T` loop of dmpc_soft_bound.m:115-148 /
 * DMPC::solveParallelDMPCv2 (dmpc/cpp/dmpc.cpp:1656-1686) incl. initDMPC at k = 1, the table
 * swap l = new_l and the ReachedGoal.m test.  Host pointers.
 * K_T_max = number of history COLUMNS, the initial state (k = 1, initDMPC) included: at most K_T_max - 1 solves per agent.
 * `for k = 1:K_T` of dmpc_soft_bound.m:115 is K_T_max = K_T; `while ~reached_goal && k < max_K` of test/failure_rate.m:99
 * records max_K - 1 columns: pass K_T_max = max_K - 1.  ReachedGoal.m is evaluated on every column, the first included.
 * A scene stops at the first step where any agent's status is not exactly DMPC_ST_SOLVED (infeasible, collided, out of
 * the workspace): stricter than failure_rate.m, which only aborts on `~feasible` and -- solveSoftDMPCbound keeping
 * feasible = 1 on outbound -- carries on with a partially updated table after an outbound first step (:112-126).
 * A QP that is infeasible without any collision row is reported DMPC_ST_INFEAS at once: the reference retries it up to 30
 * times with quadprog's ConstraintTolerance doubled each time (solveSoftDMPCbound.m:140-146) and may end up accepting a
 * bound-violating point, typically reported `outbound` instead.
 * po,pf: [S][N][3]; pk,vk,ak: [S][N][K_T_max][3] (written up to K_T_used[s]; the columns a scene never reached stay zero, like the preallocated arrays of the reference) or all three NULL: the histories
 * then only stay on the device (for dmpc_postcheck) and the 3 x S*N*K_T_max*24-byte download is skipped;
 * K_T_used[S]: number of MPC steps taken per scene; scene_status[S]: OR of agent status bits at
 * the step where the scene stopped; DMPC_ST_SOLVED | DMPC_ST_REACHED = every agent within error_tol of its goal
 * (ReachedGoal.m), DMPC_ST_SOLVED alone = ran to K_T_max without reaching (failed_goal, failure_rate.m:131-134).
 * Batches of 32 or more scenes are run in parts (2; 4 from 128 scenes on) on internal contexts of their own (a HIP stream and a
 * helper host thread each for the duration of the call) so that the slow tail of one part overlaps the others; scenes are independent,
 * results do not depend on the split.  On a DMPC_DEVICE_ALL context the agents of every scene are sharded over the GPUs instead
 * (see dmpc_create), and batches of 64 or more scenes run as two such groups side by side. */
DMPC_API int dmpc_transition(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max,
                    double error_tol, double *pk, double *vk, double *ak, int32_t *K_T_used,
                    int32_t *scene_status);

/* Multi-GPU: the agents of every scene sharded over the GPUs of one node, ONE PROCESS (rank) PER GPU, each with its own
 * context.  Replaces the thread clusters of DMPC::solveParallelDMPCv2 (dmpc/cpp/dmpc.cpp:1570-1686): contiguous agent ranges,
 * N/G each, the first N mod G one more (:1600-1625; dmpc_partition), every cluster reading the previous predictions of all
 * agents (`prev_obs = obs` after the join, :1671-1681; `l = new_l`, dmpc/matlab/dmpc_soft_bound.m:146).  The join is one
 * RCCL all-gather over xGMI per MPC step on the context's stream, the termination test (ReachedGoal.m / the abort of
 * test/failure_rate.m:112-125) a second tiny one grouped with it.
 *   dmpc_comm_unique_id   rank 0: a 128-byte id (ncclGetUniqueId) to hand to every rank by any out-of-band means
 *   dmpc_comm_init        every rank: joins the communicator (ncclCommInitRank on the context's device); nranks = 1 works
 *   dmpc_comm_destroy     leaves it (dmpc_destroy does this too)
 * A context without a communicator behaves as rank 0 of 1 (the exchange is a device copy).
 * Table layout for N agents on G ranks: lT[G][S][3K][Cmax], Cmax = ceil(N/G); the last column of the chunks of the short
 * ranks is padding that is never read. */
DMPC_API int dmpc_partition(int N, int G, int rank, int32_t *lo, int32_t *count, int32_t *cmax);
DMPC_API int dmpc_comm_unique_id(char *id128);
DMPC_API int dmpc_comm_init(dmpc_ctx *ctx, const char *id128, int nranks, int rank);
DMPC_API int dmpc_comm_destroy(dmpc_ctx *ctx);
/* ranks of the context's communicator as RCCL itself counts them (ncclCommCount); the group size of a DMPC_DEVICE_ALL context; 1 otherwise */
DMPC_API int dmpc_comm_size(const dmpc_ctx *ctx);

/* One MPC step of this rank's agents + the exchange (cluster_solvev2 + the join, dmpc.cpp:1656-1686,1792-1841).  Device
 * pointers, asynchronous on `stream`: x_p, x_v, x_a, pf: [S][count][3] and p_out, v_out, a_out: [S][count][3K] of this
 * rank's `count` agents (dmpc_partition); lT, lT_next: the whole table [G][S][3K][Cmax] before / after the step -- every
 * rank's new predictions arrive in its slot of lT_next by the all-gather.  status [S][count], info [S][count][8] or NULL. */
DMPC_API int dmpc_step_sharded_device(dmpc_ctx *ctx, int S, int N, const double *lT, const double *x_p, const double *x_v,
                             const double *x_a, const double *pf, double *p_out, double *v_out, double *a_out,
                             double *lT_next, int32_t *status, int32_t *info, void *stream);

/* The whole transition (dmpc_transition) sharded: every rank passes the SAME start/goal sets po, pf [S][N][3] (host) and gets
 * the histories pk, vk, ak [S][count][K_T_max][3] of ITS agents (or NULL), and the same K_T_used[S] / scene_status[S] as the
 * other ranks (a scene stops on every rank at the step where any agent of any rank fails or all agents reached their
 * goals: the rule of dmpc_transition).  Results do not depend on the number of ranks (bit for bit). */
DMPC_API int dmpc_transition_sharded(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                            double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status);

/* The same, followed by ONE all-gather per history array: every rank then also holds the scene-wide histories [S][N][K_T_max][3]
 * on its device, which is what dmpc_postcheck(.., pk = NULL, ..) checks -- the post-checks of test/failure_rate.m:136-195 after a
 * sharded transition, on any rank (pk, vk, ak still return the rank's own agents).  A DMPC_DEVICE_ALL context does this inside
 * dmpc_transition. */
DMPC_API int dmpc_transition_sharded_gather(dmpc_ctx *ctx, int S, int N, const double *po, const double *pf, int K_T_max, double error_tol,
                                   double *pk, double *vk, double *ak, int32_t *K_T_used, int32_t *scene_status);

/* f-1: the post-checks the reference runs after every transition -- test/failure_rate.m:136-195 (identical
 * blocks: test/comp_kctr.m:141-205, test/comp_hardsoft2.m:140-204, dmpc/matlab/dmpc_soft_bound.m:152-190):
 *   r_factor = min_{i,k} min(amax/|a_k|, vmax/|v_k|), h_scaled = h/sqrt(r_factor)       (:138-146)
 *   rescale  a_k *= r_factor; v,p re-integrated with h_scaled                           (:156-162)
 *   p(t) = spline(tk, pk, 0:Ts:T) (MATLAB not-a-knot cubic), T = (K_T_used-1) h_scaled  (:149-167)
 *   violation = any pair with |E1 (p_i - p_j)| < rmin - 0.05 at any sample              (:170-181)
 *   totdist = sum of sample-to-sample path lengths over all agents                      (:183)
 *   traj_time = Ts * max_i (1 + last sample index with |p_i - pf_i| >= 0.05)            (:186-194)
 * pk, vk, ak: [S][N][KT_alloc][3] un-rescaled histories exactly as dmpc_transition returns them, K_T_used[S]
 * valid columns per scene; pass pk = vk = ak = NULL to use the histories the last dmpc_transition of this
 * context left resident on the device.  Inputs are not modified.  Any output pointer may be NULL.
 * scene_mask (optional, [S]): 0 skips a scene (the reference only post-checks trials that stayed feasible and
 * reached their goals, failure_rate.m:136); skipped scenes report NaN / 0.  A checked scene whose histories are
 * all zero is an error (MATLAB: h_scaled = 0 makes tk empty and spline() fails).
 * p_interp (optional): [S][N][ns_alloc][3] interpolated positions (samples >= n_samples[s] are zero).
 * Any number of agents per scene: up to 256 the pairwise check is the literal all-pairs search; larger scenes bin the agents of every
 * sample into a uniform cell grid (cell = 2 rmin in the metric of the check), test the 27-neighbourhood with the same fp64 expression and
 * fall back to a tiled all-pairs pass for scenes without any pair that close -- min_dist and violation are exact either way.
 * After dmpc_transition on a DMPC_DEVICE_ALL context, or dmpc_transition_sharded_gather, the resident histories are the scene-wide ones. */
DMPC_API int dmpc_postcheck(dmpc_ctx *ctx, int S, int N, int KT_alloc, const int32_t *K_T_used, const int32_t *scene_mask,
                   const double *pk, const double *vk, const double *ak, const double *pf, double vmax, double amax, double Ts,
                   double *r_factor, double *h_scaled, int32_t *n_samples, double *min_dist, int32_t *violation,
                   double *totdist, double *traj_time, double *p_interp, int ns_alloc);

/* f-3: dense collision rows behind the CollConstr / AddCollConstr helpers named in the north star.  All of them
 * compute, per neighbour j (E1 = diag(1,1,1/c), E2 = E1^order; the ORDER is the context's, dmpc_params.order: 2, or 4 on a context of an
 * all-neighbour variant -- the helpers are generic in it, CollConstrSoftDMPC.m:16-21, and test/comp_test_ellipconstr.m:158 sets 4):
 *     dist = |E1 (p - p_j)|_order,  diff = E2 (p - p_j).^(order-1),  pd = dist^(order-1),
 *     r = pd (rmin - dist + diff.p/pd) - diff.a0,   Ain(row,:) = -diff_mat * A = -(diff . A(3 k_blk + 1..3, :)),  bin(row) = -r
 * (`dist` below returns pd = prev_dist, what CollConstrSoftDMPC.m:19 hands back: the distance itself for order 2)
 * replacing (file:line)
 *     dec-iSCP/CollConstr.m:1-24                (k_cmp = k-1, k_blk = k-2, a0 = po; all obstacles of `l`)
 *     dmpc/matlab/CollConstrSoftDMPC.m:1-32     (k_cmp = k_blk = k-1, a0 = A_initp(3k-2:3k,:)[po;vo]; `violation` mask)
 *     dmpc/matlab/CollConstrSoftDMPC2.m:8       (k_blk = k-2)     CollConstrHardDMPC.m:19 (all j != n, dist < 1)
 *     CollConstrHardDMPCOnDemand.m, CollConstrEllipDMPC.m, CollConstrSoftDMPCall.m
 * l: [N_obs][K][3] (== MATLAB l(3,K,N_obs)); sel: the n_sel 0-based obstacle indices to build rows for, in output
 * order; k_cmp / k_blk: 0-based horizon column compared / 3-row block of A used; A: a_rows x ncols with element
 * (i,j) at A[i*a_rs + j*a_cs] (MATLAB column-major: a_rs = 1, a_cs = a_rows); Ain: n_sel x ncols with strides
 * o_rs / o_cs; bin, dist (optional): [n_sel].  The *_device form takes device pointers and a HIP stream. */
DMPC_API int dmpc_coll_rows(dmpc_ctx *ctx, int K, int N_obs, int n_sel, const int32_t *sel, const double *l, int k_cmp, int k_blk,
                   const double *p, const double *a0, double rmin, double c, const double *A, int a_rows, int ncols,
                   int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs, double *bin, double *dist);
DMPC_API int dmpc_coll_rows_device(dmpc_ctx *ctx, int K, int n_sel, const int32_t *d_sel, const double *d_l, int k_cmp, int k_blk,
                          const double *p, const double *a0, double rmin, double c, const double *d_A, int64_t a_rs,
                          int64_t a_cs, int ncols, double *d_Ain, int64_t o_rs, int64_t o_cs, double *d_bin, double *d_dist,
                          void *stream);

/* dense form of structured rows as dmpc_rows_one returns them: Ain(r,:) = -(xi_r . A(3 kc_r - 2 .. 3 kc_r, :)), kc 1-based
 * (`Ain_total(idx,:) = -diff_mat*Ain`, CollConstrSoftDMPC.m:24-27).  xi [nr][3], kc [nr]; A / Ain strided as above. */
DMPC_API int dmpc_rows_dense(dmpc_ctx *ctx, int nr, const double *xi, const int32_t *kc, const double *A, int a_rows, int ncols,
                    int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs);

/* cup-SCP/AddCollConstr.m:1-31: the K N(N-1)/2 pairwise rows (pairs i<j in order, k fastest) of the coupled QP
 *     r = dist (rmin - dist) + diff.(p_i,k - p_j,k) - diff.(po_i - po_j)
 *     Ain(row,:) = -(diff . A(blk(i,k),:) - diff . A(blk(j,k),:)),  blk(i,k) = rows 3K(i-1)+3(k-1)+1..3;  bin = -r
 * p: [N][K][3] (== MATLAB p(3,K,N)); po: [N][3]; A: 3KN x ncols (strided as above); Ain: K N(N-1)/2 x ncols. */
DMPC_API int dmpc_add_coll_constr(dmpc_ctx *ctx, int K, int N, const double *p, const double *po, double rmin, double c,
                         const double *A, int ncols, int64_t a_rs, int64_t a_cs, double *Ain, int64_t o_rs, int64_t o_cs,
                         double *bin);
DMPC_API int dmpc_add_coll_constr_device(dmpc_ctx *ctx, int K, int N, const double *d_p, const double *d_po, double rmin, double c,
                                const double *d_A, int64_t a_rs, int64_t a_cs, int ncols, double *d_Ain, int64_t o_rs,
                                int64_t o_cs, double *d_bin, void *stream);

/* f-2: the reference's on-disk result formats (host code, no context needed; return 0 / -1 + dmpc_last_error(NULL)).
 * dmpc_trajectories2file == DMPC::trajectories2file (dmpc/cpp/dmpc.cpp:2088-2126), the text file
 * dmpc/cpp_results/read_result.m:4-44 reads: header `N N_cmd h_scaled pmin' pmax'`, then po (3 x N), pf (3 x N_cmd),
 * then the 3 x T position block of every trajectory, then the velocity blocks, then the acceleration blocks -- each
 * matrix in Eigen's default stream format (6 significant digits, columns aligned per matrix), byte for byte.
 * po [N][3], pf [N_cmd][3], pos/vel/acc [N_cmd][T][3] (== MATLAB pk(3,T,N_cmd)). */
DMPC_API int dmpc_trajectories2file(const char *path, int N, int N_cmd, int T, double h_scaled, const double *pmin,
                           const double *pmax, const double *po, const double *pf, const double *pos,
                           const double *vel, const double *acc);
/* test2file (dmpc/cpp/cluster_test.cpp:9-33; read by dmpc/cpp_results/cluster_test.m): header
 * `n_cluster n_vehicles n_trials`, the cluster sizes and vehicle counts, then one n_vehicles x n_trials block of
 * wall times per cluster size.  times [n_cluster][n_vehicles][n_trials]. */
DMPC_API int dmpc_test2file(const char *path, int n_cluster, int n_vehicles, int n_trials, const double *cluster_size,
                   const double *num_vehicles, const double *times);

/* f-2: the reference's start/goal generators for S scenes at once.
 * dmpc_random_test     == randomTest.m:1-60: N starts and, independently, N goals uniform in [pmin, pmax], each
 *                         farther than rmin (|E1 (p_i - p_j)|_2, E1 = diag(1,1,1/c)) from the earlier points of its
 *                         set; rejection sampling, <= 200000 tries per point, else the set restarts.
 * dmpc_random_exchange == randomExchange.m:1-57: starts as above with the Euclidean distance, goals = starts
 *                         permuted by the .m's draw-without-replacement rule (no agent keeps its own start).
 * po, pf: [S][N][3].  MATLAB's global rand stream cannot be reproduced: draws come from a counter-based stream
 * (splitmix64 of seed, scene, set, draw index), so a (seed, S, N, box, rmin) tuple always gives the same scenes.
 * The _device form writes [2][S][N][3] (starts, then goals) to device memory on the caller's stream. */
DMPC_API int dmpc_random_test(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                     uint64_t seed, double *po, double *pf);
DMPC_API int dmpc_random_exchange(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, uint64_t seed,
                         double *po, double *pf);
DMPC_API int dmpc_random_sets_device(dmpc_ctx *ctx, int S, int N, const double *pmin, const double *pmax, double rmin, double c,
                            uint64_t seed, int exchange, double *d_po_pf, void *stream);

/* Standalone forms of the path's small helpers (fused into the step kernels inside the solvers), for callers that
 * invoke them on their own as the reference's scripts do.  Host pointers, synchronous.
 *   dmpc_prop_state   propStatedmpc.m:1-8 (A_initp given): p = A_p a + A_initp [po;vo], v = A_v a + 1 (x) vo  -> pass
 *                     off_v = vo; dec-iSCP/propState.m:1-10 (A_initp = NULL): new_p + repmat(po) -> pass off_p = po.
 *                     A_p, A_v: [n_rows][n_cols] row-major, A_initp: [n_rows][6] or NULL; off_p/off_v: 3-vectors tiled
 *                     over the rows (or NULL).
 *   dmpc_is_inbounds  is_inbounds.m:1-6 on npts points [npts][3] (5 cm tolerance)
 *   dmpc_reached_goal ReachedGoal.m:1-11: max_i |p_i - pf_i| < error_tol, p, pf [N][3]
 *   dmpc_max_deviation maxDeviation.m:1-11 (the stopping measure of solveDMPC.m:69): p, prev_p [K_cols][3] (== MATLAB 3 x K_cols); the
 *                     largest per-step distance over the steps the .m looks at -- `K = length(p)/3` of the MATRIX p is max(3, K_cols)/3,
 *                     i.e. the first 5 of 15 horizon steps, restated as written */
DMPC_API int dmpc_prop_state(dmpc_ctx *ctx, int n_rows, int n_cols, const double *A_p, const double *A_v, const double *A_initp,
                    const double *po, const double *vo, const double *off_p, const double *off_v, const double *a,
                    double *p, double *v);
DMPC_API int dmpc_is_inbounds(dmpc_ctx *ctx, int npts, const double *p, const double *pmin, const double *pmax, int32_t *inbounds);
DMPC_API int dmpc_reached_goal(dmpc_ctx *ctx, int N, const double *p, const double *pf, double error_tol, int32_t *reached);
DMPC_API int dmpc_max_deviation(dmpc_ctx *ctx, int K_cols, const double *p, const double *prev_p, double *tol_out);

/* Agent-steps LAUNCHED by this context so far (incl. the half of a split dmpc_transition batch that runs on the internal
 * second context).  An upper bound of the QPs actually solved: agents of scenes that already stopped are skipped on the
 * device, and agents the scan certifies infeasible never enter the solver; per-agent outcomes are in status[]. */
DMPC_API int64_t dmpc_solve_count(const dmpc_ctx *ctx);

/* Roofline instrumentation: with dmpc_profile(ctx,1) every step-kernel launch is bracketed by HIP
 * events on the stream it is launched on; dmpc_profile_read drains them and returns the average
 * duration (ms) of the step's kernels and the step count since the previous read. */
DMPC_API int dmpc_profile(dmpc_ctx *ctx, int enable);
DMPC_API int dmpc_profile_read(dmpc_ctx *ctx, double *avg_ms, int64_t *n_launches);
/* same, split into the solve kernel(s) (dmpc_solve_kernel, the dominant kernel) and scan + ordering */
DMPC_API int dmpc_profile_read2(dmpc_ctx *ctx, double *solve_avg_ms, double *scan_avg_ms, int64_t *n_steps);
/* the solve kernel the context's last MPC step launched for the bulk of its agents, as a profiler names it (e.g. "dmpc_rsolve_persist_kernel",
 * "dmpc_solve_persist_kernel<true, 56, 48, double>"): the launch form is picked per step from the variant and the depth of the launch, and whoever
 * reports the measured duration of "the dominant kernel" (bench.py's roofline block) names the kernel from here, not from a copy of that rule.
 * The string lives in the context; empty before the first step.  (ABI revision 7) */
DMPC_API const char *dmpc_last_solve_kernel(const dmpc_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DMPC_HIP_H */
