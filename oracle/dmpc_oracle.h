/*
 * dmpc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's per-agent DMPC horizon QP
 * (carlosluis/multiagent_planning, dmpc/matlab/solve*DMPC*.m + helpers) used ONLY by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * reported baseline.  Nothing under multiagent_planning_amd/ may link or call it.
 *
 * The reference's arithmetic for the solve itself lives in MATLAB `quadprog`
 * (Optimization Toolbox, R2018-era, not version-pinned, not in /root/reference) and, for
 * dmpc/cpp, in jrl-umi3218/eigen-quadprog (QuadProgDense, dmpc/cpp/dmpc.h:13).  The QPs are
 * strictly convex (H = 2(A'QA + Delta'S Delta + I [+ EPS]) > 0, solveSoftDMPCbound.m:98), so
 * the minimiser is unique; this oracle assembles the *literal* dense QP exactly as the .m
 * files do (slack variables, +-Lambda rows, bounds) and solves it with a dense
 * Goldfarb-Idnani dual active-set method restated from the published algorithm
 * (Goldfarb & Idnani, Math. Programming 27 (1983) 1-33 -- the method eigen-quadprog
 * implements).  It is pinned against the MATLAB/quadprog golden records in tests/golden/.
 */
#ifndef DMPC_ORACLE_H
#define DMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* solver variants == reference function names (dmpc/matlab/) */
enum {
    ORC_BOUND = 0,      /* solveSoftDMPCbound.m        */
    ORC_BOUND2 = 1,     /* solveSoftDMPCbound2.m       */
    ORC_ALL3 = 2,       /* solveSoftDMPCall.m          */
    ORC_HARD = 3,       /* solveHardDMPC.m             */
    ORC_ONDEMAND = 4,   /* solveHardDMPCOnDemand.m     */
    ORC_ELLIP = 5,      /* solveEllipDMPC.m            */
    ORC_SOFTALL = 6,    /* solveSoftDMPC.m             */
    ORC_REPAIR = 7,     /* solveSoftDMPCrepair.m       */
    ORC_CPP = 8,        /* dmpc/cpp DMPC::solveQPv2, _k_factor = 0  (dmpc.cpp:803-1287) */
    ORC_CPP2 = 9,       /* dmpc/cpp DMPC::solveQPv2, _k_factor = -1                      */
    ORC_CPP1 = 10,      /* dmpc/cpp DMPC::solveQP (the first version: dmpc.cpp:554-801; rows for ALL N-1 neighbours, check_collisions :378-396, build_collconstraint :450-498) */
    ORC_SOFTALL_C = 11, /* solveSoftDMPC_c.m (solveSoftDMPC with slack penalties -1e4 (K/k)^2, 1e6 (K/k)^2, :60-63; test/comp_confidence.m:184) */
    ORC_SCP = 12        /* solveDMPC.m (the legacy spherical SCP loop, :17-72; dmpc/matlab/dmpc.m:79, test/success_test_dmpc.m:80) */
};

/* status bits returned per agent */
enum {
    ORC_ST_SOLVED = 1,    /* p,v,a outputs valid                                   */
    ORC_ST_OUTBOUND = 2,  /* first predicted position outside box +-5cm            */
    ORC_ST_COLL = 4,      /* already collided at horizon step 1 (no outputs)       */
    ORC_ST_INFEAS = 8     /* QP (after the retry ladder) infeasible (no outputs)   */
};

typedef struct {
    int K;          /* horizon length k_hor (15)                         */
    int variant;    /* ORC_*                                             */
    int order;      /* ellipsoid order: 2; 4 for the all-neighbour variants (softall, softall_c, ellip, repair, cpp1) */
    int max_tries;  /* <=0: reference default (30; repair: 10; cpp: 21)  */
    double h;       /* time step                                         */
    double rmin;    /* collision radius                                  */
    double c;       /* E = diag(1,1,c)                                   */
    double alim;    /* acceleration limit                                */
    double Q1, S1;  /* collision-case weights                            */
    double term;    /* linear slack penalty (negative)                   */
    double pmin[3], pmax[3];
    double Qfar, Qnear, Sfree; /* collision-free cost cases; 0 = HEAD constants 1000 / 10000 / 10 (solveSoftDMPCbound.m:44-52) */
    double tol;     /* ORC_SCP only: `tol` of solveDMPC.m:1,17 (stop when maxDeviation(p, prev_p) <= tol)                   */
} orc_params;

/* info[] layout (8 ints per agent) */
enum { ORC_I_VIOLK = 0, ORC_I_NV = 1, ORC_I_TRIES = 2, ORC_I_CASE = 3, ORC_I_ITERS = 4,
       ORC_I_NSLACK = 5, ORC_I_NACTIVE = 6, ORC_I_NROWS = 7, ORC_INFO_LEN = 8 };

/* a1-a3: model matrices, row-major. Lambda,Av,Delta: 3K x 3K; A0: 3K x 6 */
int orc_model_matrices(double h, int K, double *Lambda, double *Av, double *A0, double *Delta);

/* a4: initDMPC.m, p/v/a: 3K each ([x1 y1 z1 x2 ...]) */
int orc_init_one(const double po[3], const double pf[3], double h, int K, double *p, double *v, double *a);

/* a7/a8: one agent, one MPC step. l: N x 3K row-major table; n: 0-based agent index.
 * p,v,a: 3K outputs (untouched unless SOLVED). obj: literal QP objective 1/2 x'Hx + f'x
 * (including slack terms) at the solution. Returns status bits. */
int orc_solve_one(const orc_params *prm, int N, int n, const double *l, const double po[3],
                  const double vo[3], const double ao[3], const double pf[3], double *p, double *v,
                  double *a, int *info, double *obj);

/* a11: all N agents of one MPC step; nthreads contiguous clusters as dmpc.cpp:1600-1625.
 * x_p,x_v,x_a,pf: N x 3; p,v,a: N x 3K; status: N; info: N x 8; obj: N (may be NULL). */
int orc_step(const orc_params *prm, int N, const double *l, const double *x_p, const double *x_v,
             const double *x_a, const double *pf, double *p, double *v, double *a, int *status,
             int *info, double *obj, int nthreads);

/* evaluate the literal QP of agent n at a candidate acceleration vector `acc` (3K): computes
 * the optimal slack for that acc, returns objective and max constraint violation (Tier-B
 * comparison against quadprog records). */
int orc_eval_one(const orc_params *prm, int N, int n, const double *l, const double po[3],
                 const double vo[3], const double ao[3], const double pf[3], const double *acc,
                 double *obj, double *maxviol);

/* a5/a6 standalone: collision rows of agent n exactly as the solver variant builds them (dense) */
int orc_rows_one(const orc_params *prm, int N, int n, const double *l, const double po[3], const double vo[3],
                 int max_rows, double *G, double *b, double *dist, int *nrows, int *viol_k, int *status);

/* generic dense strictly-convex QP: min 1/2 x'Gx + g'x  s.t. C x <= d (m rows), used by tests.
 * lam: m multipliers. returns 0 ok, 1 infeasible, <0 error */
int orc_qp_dense(int n, int m, const double *G, const double *g, const double *C, const double *d,
                 double *x, double *lam, int *iters);

/* the literal dense QP of agent n at retry-ladder level `level` (min 1/2 x'Hx + f'x s.t. Cx <= d, x = [a; eps]) for the
 * solver-independent certificates of tests/ (KKT by NNLS, infeasibility by an LP).  Call with H == NULL for the sizes. */
int orc_assemble_one(const orc_params *prm, int N, int n, const double *l, const double po[3], const double vo[3],
                     const double ao[3], const double pf[3], int level, int *n_out, int *m_out, int *ncoll_out,
                     double *H, double *f, double *C, double *d);

/* S independent scenes of N agents, scene-parallel over nthreads host threads (the CPU baseline of bench.py) */
int orc_step_scenes(const orc_params *prm, int S, int N, const double *l, const double *x_p, const double *x_v, const double *x_a,
                    const double *pf, double *p, double *v, double *a, int *status, int *info, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
