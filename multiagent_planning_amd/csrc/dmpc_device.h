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
__host__ __device__ inline int tcol(int j) { const int g = j >> 3; return 8 * g * (4 * g + 5) + (8 * g + 9) * (j - 8 * g); }
// doubles of T for a given working-set capacity (columns up to the next multiple of 8).  Unconditional wave reads run
// past the last column by up to 63 doubles; their values are masked, and the vectors that follow T in the per-agent LDS
// block (>= 4 x 48 doubles) are what they touch.
__host__ __device__ inline int t_doubles(int qcap) { return tcol((qcap + 7) & ~7); }
constexpr int ITER_CAP = 4000;

enum { VAR_BOUND = 0, VAR_BOUND2 = 1, VAR_ALL3 = 2, VAR_HARD = 3, VAR_ONDEMAND = 4, VAR_ELLIP = 5, VAR_SOFTALL = 6, VAR_REPAIR = 7,
       VAR_CPP = 8, VAR_CPP2 = 9 /* dmpc/cpp solveQPv2 with _k_factor 0 / -1 */ };
enum { ST_SOLVED = 1, ST_OUTBOUND = 2, ST_COLL = 4, ST_INFEAS = 8, ST_CAPACITY = 16, ST_ITERCAP = 32,
       ST_QOVER = 64 /* internal: tier-1 working set overflowed, tier 2 re-solves */ };

struct StepParams {
    int variant, S, G, C, g_local, nrmax, max_tries, c_first, c_count, qcap;
    double h, rmin, e1z, e2z, alim, Q1, S1, term;
    double pmin[3], pmax[3];
    double hsum[3];         // per cost case: sum of |H1(i,j)| (bound of the cost over the acceleration box, dual-bound certificate)
    const double *tables;   // [3 cost cases][H1^-1 | H1^-1 L' | L H1^-1 L'][15*15]
    const double *lT;       // [G][S][3K][C]
    const double *x_p, *x_v, *x_a, *pf;   // [S][c_count][3]  (agents c_first .. c_first+c_count-1 of chunk g_local)
    double *p_out, *v_out, *a_out;        // [S][c_count][3K]
    double *lT_next;                      // [S][3K][C] or null
    int *status, *info;
    double *rowbuf;         // per-agent collision-row scratch [S*c_count][nrmax*(soft?7:4)] doubles (global, L2-resident)
    int *rowkc;             // [S*c_count][nrmax] constrained horizon step of each row
    int *hdr;               // [S*c_count][8] scan -> solve hand-off (row count, branch record)
    int only_flagged, qover_bit;
    int no_prune;           // 1: keep every row the reference builds (dmpc_rows_one); 0: exact pruning
    const double *bbox;     // [G][S][6][C] horizon bounding boxes (bbox_kernel) or null: neighbour culling in the scan
    const int *order;       // solve-phase launch order (agent ids, heaviest first) or null
    int nbr_cap;            // scan: capacity of the LDS neighbour list (0: no list)
    const int *scene_done;  // [S] or null: scenes of a transition that already stopped (reached their goals / failed): skipped
    int *counter;           // persistent solve kernel: queue head (zeroed before the launch)
    int lds_per_wave;       // persistent solve kernel: bytes of LDS per wave (after the shared tables)
    int *flag_count, *flag_list;   // tier 1 -> tier 2: number / ids of the agents whose working set overflowed (or null)
    double *dbg;            // optional per-iteration trace of agent dbg_agent (development aid)
    int dbg_agent, dbg_cap;
};

constexpr int SCAN_CAND_CAP = 1024;   // (neighbour, step) candidates buffered per flush of the hard-row scan

// bytes of dynamic LDS the step kernel carves for a given row capacity
inline size_t step_lds_bytes(int nrmax, bool soft, int qcap, int phase, int nbr_cap = 0)
{
    if (phase == 0) return 96 * 8 + (SCAN_CAND_CAP + (size_t)nbr_cap) * 4;   // own prediction, unconstrained minimiser + candidate list of the all-k (hard) scan + neighbour list
    size_t dbl = (size_t)t_doubles(qcap) + 676 + 4 * 48 + 2 * 64 + 6 * (size_t)qcap + (soft ? (size_t)nrmax : 0);
    size_t bytes = dbl * 8 + 3 * (size_t)qcap * 4 + (size_t)nrmax;   // + slot ints + row flags (bytes)
    return (bytes + 15) & ~(size_t)15;
}
// persistent solve kernel: the tables are shared by the workgroup (3 cost cases x 676 doubles at the front)
constexpr size_t PERSIST_TABLE_BYTES = 3 * 676 * 8;
inline size_t persist_wave_bytes(int nrmax, bool soft, int qcap) { return step_lds_bytes(nrmax, soft, qcap, 1) - 676 * 8; }

}  // namespace dmpc
