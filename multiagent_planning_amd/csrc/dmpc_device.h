// dmpc_device.h -- constants and the kernel parameter block shared by dmpc_kernels.hip and dmpc_api.hip
#pragma once
#include <stdint.h>

namespace dmpc {

constexpr int K = 15;         // horizon k_hor (dmpc_soft_bound.m:13)
constexpr int N3 = 3 * K;     // stacked acceleration / position vector length
constexpr int QMAX = 64;      // working-set capacity (slots of the inverse factor T) == wave size
// Inverse-factor storage: upper-triangular T, column-major; the 8 columns of group g = j/8 are zero-padded to 8(g+1)
// rows and stored with a stride of 8(g+1)+1 doubles.  The ODD stride is for the LDS banks: in the transposed product
// lane j walks down column j, and with the natural stride 8(g+1) all columns of a group start on the same few banks
// (measured: 40 % of the LDS cycles of the solve kernel were bank-conflict replays).
__host__ __device__ constexpr int tcol(int j) { return 8 * (j >> 3) * (4 * (j >> 3) + 5) + (8 * (j >> 3) + 9) * (j - 8 * (j >> 3)); }
// doubles of T for a given working-set capacity (columns up to the next multiple of 8).  Unconditional wave reads run
// past the last column by up to 63 doubles; their values are masked, and the vectors that follow T in the per-agent LDS
// block (>= 4 x 48 doubles) are what they touch.
__host__ __device__ constexpr int t_doubles(int qcap) { return tcol((qcap + 7) & ~7); }
#ifndef DMPC_HARD_TS
#define DMPC_HARD_TS 16
#endif
#ifndef DMPC_HARD_PW
#define DMPC_HARD_PW 12
#endif
constexpr int HARD_PW = DMPC_HARD_PW;   // persistent waves per CU the slack-free solve kernel is compiled for (12 = three per SIMD at 168 registers per lane)
constexpr int SOFT_TS = 48;             // slack variants, 56-slot tier (large scenes): own columns of T per wave; columns 48 .. 55 from the workgroup's pool
constexpr int HARD_TS = DMPC_HARD_TS;   // slack-free persistent solve: columns of T in a wave's own LDS block (split T, dmpc_solve.hip); the other 48 - HARD_TS come from the workgroup's pool
constexpr int ITER_CAP = 4000;
constexpr int CRASH_MIN_DEFAULT = 4;
constexpr int NBR_PARTS = 16;   // nbr_kernel: the neighbours of a scene are split over this many waves per block of 64 agents (N / 64 waves alone do not fill the chip)
constexpr int NBOX_NM = 20;     // floats per neighbour in the neighbour-major copy of the segment boxes (18 used; five 16-byte scalar loads)

enum { VAR_BOUND = 0, VAR_BOUND2 = 1, VAR_ALL3 = 2, VAR_HARD = 3, VAR_ONDEMAND = 4, VAR_ELLIP = 5, VAR_SOFTALL = 6, VAR_REPAIR = 7,
       VAR_CPP = 8, VAR_CPP2 = 9 /* dmpc/cpp solveQPv2 with _k_factor 0 / -1 */, VAR_CPP1 = 10 /* dmpc/cpp solveQP (first version, dmpc.cpp:554-801) */,
       VAR_SOFTALL_C = 11 /* solveSoftDMPC_c.m */, VAR_SCP = 12 /* solveDMPC.m: the SCP loop (dmpc_scp_kernel) */ };
enum { ST_SOLVED = 1, ST_OUTBOUND = 2, ST_COLL = 4, ST_INFEAS = 8, ST_CAPACITY = 16, ST_ITERCAP = 32,
       ST_QOVER = 64 /* internal: tier-1 working set overflowed, tier 2 re-solves */ };

struct StepParams {
    int variant, S, G, C, g_local, nrmax, max_tries, c_first, c_count, qcap;
    int ell_order;          // super-ellipsoid order: 2, or 4 for the all-neighbour variants (softall, ellip, repair, cpp1)
    double h, rmin, e1z, e2z, alim, Q1, S1, term;
    double Qfar, Qnear, Sfree;   // weights of the collision-free cost cases (HEAD: 1000, 10000, 10)
    double pmin[3], pmax[3];
    double hsum[3];         // per cost case: sum of |H1(i,j)| (bound of the cost over the acceleration box, dual-bound certificate)
    const double *tables;   // [3 cost cases][30x30 Gram table G] + [15x15 Lambda' table]  (TAB_DOUBLES; see dmpc_solve.hip)
    const double *lT;       // [G][S][3K][C]; mixed precision: the scan reads it as a float table (same layout)
    const double *own_prev; // mixed precision: fp64 predictions of chunk g_local [S][3K][C] (the solve's fallback for unsolved agents), else null
    const double *x_p, *x_v, *x_a, *pf;   // [S][c_count][3]  (agents c_first .. c_first+c_count-1 of chunk g_local)
    double *p_out, *v_out, *a_out;        // [S][c_count][3K]
    double *lT_next;                      // [S][3K][C] or null
    int *status, *info;
    double *rowbuf;         // per-agent collision-row scratch [S*c_count][nrmax*(soft?7:4)] doubles (global, L2-resident)
    int *rowkc;             // [S*c_count][nrmax] constrained horizon step of each row
    int *hdr;               // [S*c_count][8] scan -> solve hand-off (row count, branch record)
    int only_flagged, qover_bit;
    int no_prune;           // 1: keep every row the reference builds (dmpc_rows_one); 0: exact pruning
    const int *order;       // solve-phase launch order (agent ids, heaviest first) or null
    int nbr_cap;            // capacity of an agent's neighbour list (ints)
    int *nbr_list;          // [S*c_count][nbr_cap] neighbours that can come close, in increasing index order (nbr_kernel), or null
    const void *lrow;       // neighbour-major copy of the table the scan reads (table_nbrmajor_kernel), with the lists; or null
    const int *nbr_cnt;     // [S*c_count][NBR_PARTS] entries of each piece of a list; -1: did not fit (the scan walks the whole table)
    // closed loops, tiny launches: the step after the solve (state advance, history column, scene verdict: post_step_kernel) done by
    // the solve kernel itself -- each wave for its agent, the last wave of a scene (a counter) for the verdict; post_on = 0: off
    int post_on, post_KT, post_k;
    double post_tol;
    double *post_xp, *post_xv, *post_xa, *post_pk, *post_vk, *post_ak;
    int *post_flags, *post_done;
    unsigned long long *post_max;   // [S] bit pattern of the largest goal distance so far (non-negative doubles order like integers)
    int *post_or, *post_cnt;        // [S] OR of the status words, number of agents accounted for
    const int *scene_done;  // [S] or null: scenes of a transition that already stopped (reached their goals / failed): skipped
    int *counter;           // persistent solve kernel: queue head (zeroed before the launch)
    int *zero4;             // scan kernel: four ints its first workgroup sets to zero (the queue heads / tier-2 count / live bound of this step: no memset launch), or null
    int lds_per_wave;       // persistent solve kernel: bytes of LDS per wave (after the shared tables)
    int *flag_count, *flag_list;   // tier 1 -> tier 2: number / ids of the agents whose working set overflowed (or null)
    double *dbg;            // optional per-iteration trace of agent dbg_agent (development aid)
    int dbg_agent, dbg_cap;
    int short_from;         // unequal clusters: chunks r >= short_from hold C-1 agents (last column = padding); 0: all chunks hold C
    int crash_min;          // crash start of the acceleration bounds when at least this many are violated at the unconstrained minimiser (0: never)
    int pivot_explore;      // development builds (DMPC_PIVOT_EXPLORE): extra pivot-weight multipliers, see dmpc_solve.hip
    int fast_exit;          // scan: finish the agents whose unconstrained minimiser is feasible (they never enter the solve queue)
    const int *live_bound;  // persistent solve kernel: queue positions from *live_bound on hold agents the scan finished (order_kernel), or null
    int no_level_check;     // development option no_level_check: solveSoftDMPCall without the slack-free feasibility pass per ladder level (A/B runs, tests)
    double dep_tol_f32;     // fp32-factor kernels: dependence threshold on delta / s_pp (development option f32_dep_exp: 10^-n)
    int queue_chunk;        // persistent solve kernel: adjacent queue positions a ticket of the light bulk stands for (1 or 2, launch_step)
    int n_ext;              // persistent solve kernel with a split T: extensions in the workgroup's pool (behind the waves' blocks)
    int *cost_out;          // [S*c_count] or null: work estimate of this agent's solve (quarter microseconds: iterations weighted by the working-set size,
                            // certificate calls) -- the NEXT step's launch order is built from it (order_kernel; dmpc_api.hip: order hint)
    double scp_tol;         // VAR_SCP: `tol` of solveDMPC.m:1,17 (the loop stops when maxDeviation(p, prev_p) <= tol)
    int *gzero;             // scan kernel: the cell grid's counters (cell counts, largest half extents), set to zero for the NEXT step's grid_prep_kernel (no memset launch), or null
    int gzero_n;
    int no_level_skip;      // development option no_level_skip: the retry ladder does not extrapolate a failed solve's Farkas combination to higher levels (every level not
                            // certified infeasible is solved: A/B runs of the extrapolation's margin, tests/test_gpu_reduced.py)
    int rsolve_cap;         // reduced solver (dmpc_rsolve.hip): equality-constrained solves per ladder level before an agent is handed to the general solver (0: REQP_MAX; development option rsolve_cap)
    int iter_cap;           // active-set iteration cap per try (ITER_CAP; development runs lower it to measure the per-iteration cost)
};

constexpr int SCAN_CAND_CAP = 1024;   // (neighbour, step) candidates buffered per flush of the hard-row scan

// Precomputed tables (host: build_case_tables, dmpc_api.hip).  Per cost case one symmetric 30x30 Gram table over the
// index (space, step), space A = acceleration components (0..14), space W = position components (15..29):
//   G[A i][A j] = H1^-1(i,j),  G[A i][W j] = (H1^-1 L')(i,j),  G[W i][W j] = (L H1^-1 L')(i,j);
// then Lt[k][kk] = Lambda(kk,k) (15x15, the same for every case).
constexpr int TAB_CASE_DOUBLES = 900;
constexpr int TAB_L_DOUBLES = 225;
constexpr int TAB_DOUBLES = 3 * TAB_CASE_DOUBLES + TAB_L_DOUBLES;   // 2925
// behind them in the same device buffer (global memory only, never staged in LDS): per cost case the upper-triangular inverse Cholesky
// factor Tp of H1^-1 (H1^-1 = C C', Tp = C^-T; its upper triangle packed row by row: entry (i, j >= i) at i (31 - i) / 2 + j - i) -- the
// factor columns of a working set of acceleration bounds that is a prefix of the horizon (crash start of the slack variants, dmpc_solve.hip)
// Two tables per case: for the steps in rising order (a prefix 0 .. m-1 of the horizon) and in falling order (the END of the horizon,
// steps 14, 13, ..: the factor of J H1^-1 J, J the order-reversing permutation).
constexpr int TAB_TP_CASE = 120;
constexpr int TAB_TP_DOUBLES = 3 * 2 * TAB_TP_CASE;
constexpr int TAB_ALL_DOUBLES = TAB_DOUBLES + TAB_TP_DOUBLES;

// bytes of dynamic LDS of one scan wave: own prediction, unconstrained minimiser + candidate list of the all-k (hard) scan +
// neighbour list
inline size_t scan_lds_bytes() { return (96 * 8 + (size_t)SCAN_CAND_CAP * 4 + 15) & ~(size_t)15; }
// bytes of LDS of one solve wave (layout: SolveLds in dmpc_solve.hip; `persist`: the tables are shared by the workgroup)
// `tsplit` (persistent slack-free kernels, round 4): the wave's own block holds the first tsplit columns of T only; an agent whose
// working set outgrows them takes one of the workgroup's EXTENSIONS (columns tsplit .. qcap-1, ext_doubles) from a small pool
inline size_t solve_lds_bytes(int nrmax, bool soft, int qcap, bool persist, int tsplit = 0, bool f32_factor = false)
{
    size_t dbl = (f32_factor ? ((size_t)t_doubles(qcap) + 1) / 2 : (size_t)t_doubles(tsplit > 0 ? tsplit : qcap)) + 2 * 48 + 2 * 64 + 48 + 5 * (size_t)qcap + (soft ? (size_t)qcap : 0) + (size_t)qcap / 2;
    if (!persist) dbl += TAB_CASE_DOUBLES + TAB_L_DOUBLES + 1;
    if (soft) dbl += (size_t)nrmax;                 // r_eps
    const size_t bytes = dbl * 8 + (soft ? 4 * (size_t)nrmax : (((size_t)nrmax + 31) / 32) * 4);   // + row flags (soft: a byte per row + three bytes of slot map; slack-free: a bit)
    return (bytes + 15) & ~(size_t)15;
}
// persistent solve kernel: the tables at the front of the workgroup's LDS, then 16 bytes of pool header (bit mask of the free T extensions)
constexpr size_t PERSIST_TABLE_BYTES = (((size_t)TAB_DOUBLES * 8 + 15) & ~(size_t)15) + 16;
__host__ __device__ constexpr int ext_doubles(int qcap, int tsplit) { return t_doubles(qcap) - t_doubles(tsplit); }
constexpr size_t EXT_PAD_BYTES = 512;   // behind the last extension: unconditional wave reads run past a column's end (values masked)

}  // namespace dmpc
