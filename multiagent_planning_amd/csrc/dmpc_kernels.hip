// dmpc_kernels.hip -- gfx950 (MI355X, CDNA4) device code of the DMPC per-agent horizon-QP hot path.
//
// One 64-lane wavefront (= one workgroup) owns one agent for one MPC step:
//   scan   (a5)  CheckCollSoftDMPC.m:7-15 / CheckCollEllipDMPC.m:4-11 : coalesced reads of the
//                transposed prediction table lT[G][S][3K][C], lane = neighbour
//   rows   (a6)  CollConstrSoftDMPC.m:16-28 (+2/Hard/OnDemand/Ellip variants): compacted into LDS in
//                structured form (xi, kc, rhs, slack descriptor) -- 5..9 numbers per row, never dense
//   solve  (a7)  the QP of solveSoftDMPCbound.m:43-103 (and variants) by a dual active-set
//                (Goldfarb-Idnani) method in Schur-complement form: the Hessian is input-independent
//                and per-axis (H = H1 (x) I3), so H1^-1, H1^-1 L', L H1^-1 L' (15x15 each) are
//                precomputed per cost case and staged in LDS; the only per-agent matrix is the
//                upper-triangular inverse Cholesky factor T of the active-set Schur matrix
//                (T T' = (N' H^-1 N)^-1), kept in LDS and updated by column append / Givens deletes.
//                Slack variables (solveSoftDMPCbound.m:60-88) are explicit but their eps<=0 pins are
//                instantiated lazily, so the working set only holds what is really active.
//                (dmpc_solve.hip, included below)
//   prop   (a9)  propStatedmpc.m:3-4, is_inbounds.m:2-5 (a10), outputs + next table chunk.
//
// All arithmetic is fp64 (the reference is MATLAB double).  The data path has no atomics: every reduction is a
// fixed butterfly, so results are bit-reproducible and independent of how agents are sharded.  (Atomics only
// serve scheduling: the queue head of the persistent waves, the tier-2 list, the histograms of order_kernel.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "dmpc_device.h"

namespace dmpc {

// --------------------------------------------------------------------------------------------
// wave64 helpers
// --------------------------------------------------------------------------------------------
// The workgroup is ONE wavefront and the LDS executes a wave's DS instructions in issue order, so lanes
// exchanging data through LDS need no hardware barrier or counter wait -- only the compiler must not
// reorder the accesses.  (A __syncthreads() here would also drain outstanding global loads.)
#define LSYNC() asm volatile("" ::: "memory")

__device__ __forceinline__ double readlane_d(double v, int l /*uniform*/)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// Wave-wide reductions on the VALU with DPP row shifts / row broadcasts (gfx9 scan idiom): six
// steps, no LDS crossbar traffic (a ds_bpermute butterfly costs ~10x more here).  Lane 63 ends up
// with the total, which is then broadcast through an SGPR.  Fixed order => bit-reproducible.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_d(double ident, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define DMPC_WAVE_REDUCE(NAME, IDENT, OP)                                     \
    __device__ __forceinline__ double NAME(double v)                          \
    {                                                                         \
        const double id = IDENT;                                              \
        v = OP(v, dpp_d<0x111, 0xf>(id, v)); /* row_shr:1 */                  \
        v = OP(v, dpp_d<0x112, 0xf>(id, v)); /* row_shr:2 */                  \
        v = OP(v, dpp_d<0x114, 0xf>(id, v)); /* row_shr:4 */                  \
        v = OP(v, dpp_d<0x118, 0xf>(id, v)); /* row_shr:8 */                  \
        v = OP(v, dpp_d<0x142, 0xa>(id, v)); /* row_bcast:15 -> rows 1,3 */   \
        v = OP(v, dpp_d<0x143, 0xc>(id, v)); /* row_bcast:31 -> rows 2,3 */   \
        return readlane_d(v, 63);                                             \
    }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
DMPC_WAVE_REDUCE(wave_max, -INFINITY, op_max)
DMPC_WAVE_REDUCE(wave_min, INFINITY, op_min)

// value of lane + N of the same 16-lane row (row_shl DPP); lanes whose source falls off the row keep their own value
template <int N>
__device__ __forceinline__ double dpp_row_shl(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x100 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x100 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x100 + N, 0xf, 0xf, false));
}
DMPC_WAVE_REDUCE(wave_sum, 0.0, op_add)
__device__ __forceinline__ unsigned wave_or(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lanes_below(unsigned long long mask, int lane)
{
    return __popcll(mask & ((1ull << lane) - 1ull));
}

// Division of a per-lane index (< 2^31) by a wave-uniform divisor d: product with the double reciprocal (off by at most
// one), one correction step in either direction makes it exact.  Branch-free.
struct UDiv {
    int d;
    double inv;
    __device__ __forceinline__ explicit UDiv(unsigned d_) : d(d_ ? (int)d_ : 1), inv(1.0 / (double)(d_ ? d_ : 1u)) {}
    __device__ __forceinline__ void divmod(unsigned e, int &q, int &rem) const
    {
        int qq = (int)((double)e * inv);
        int rr = (int)e - qq * d;
        const int up = rr >= d ? 1 : 0, dn = rr < 0 ? 1 : 0;
        q = qq + up - dn;
        rem = rr - (up - dn) * d;
    }
};

__device__ __forceinline__ double sel3(const double *v, int ax) { return ax == 0 ? v[0] : (ax == 1 ? v[1] : v[2]); }

// constraint types
#ifndef SCAN_WAVES_PER_SIMD
#define SCAN_WAVES_PER_SIMD 4
#endif
#ifndef SCAN_WAVES_PER_WG
#define SCAN_WAVES_PER_WG 4
#endif
#ifndef QUEUE_CHUNK
#define QUEUE_CHUNK 2   // measured on the headline workload (M solves/s), tickets claimed at the end of a solve (round 3): 1: 48.9, 2: 52.0, 3: 51.4, 4: 50.5
#endif
#ifndef QUEUE_T1
#define QUEUE_T1 1
#endif
#ifndef QUEUE_T3
#define QUEUE_T3 2   // single-position tickets for the last 2 x #waves positions (1: 52.0, 2: 52.3, 4: 50.9, 8: 49.3 M solves/s)
#endif
#ifndef SOLVE_WAVES_PER_SIMD
#define SOLVE_WAVES_PER_SIMD 2
#endif
enum { TY_BOXHI = 0, TY_BOXLO = 1, TY_POSHI = 2, TY_POSLO = 3, TY_COLL = 4, TY_SLKU = 5, TY_SLKL = 6 };
// row flags
enum { RF_COLL = 1, RF_SLKU = 2, RF_SLKL = 4, RF_LIVE = 8 };

// what the scan phase addresses: its LDS vectors and the per-agent slice of the global row scratch
struct Lds {
    double *w_s, *own_s;
    double *r_xi;   // nrmax x 3
    double *r_b;
    double *r_sd, *r_st, *r_slb;  // soft variants only
    int *r_kc;
};

struct Agent {   // wave-uniform agent data
    double po[3], vo[3], ao[3], pf[3];
};

// ---- arithmetic shared by the scan's unconstrained exit and the solver (explicit contractions: both kernels must produce the
// ---- same bits for an agent whichever of them finishes it)
// cost case of solveSoftDMPCbound.m:43-58 (0 far, 1 near, 2 collision rows)
__device__ __forceinline__ int cost_case(int var, double d0, double d1, double d2 /* po - pf */, bool rows_exist)
{
    if (var == VAR_SCP) return rows_exist ? 2 : 0;   // solveDMPC.m:38-48: `isempty(Ain_total)` alone decides (Q = 1000, S = 10, or Q1, S1)
    const double dn = sqrt(fma(d2, d2, fma(d1, d1, d0 * d0)));
    const bool far = (var == VAR_ELLIP) ? (dn > 1.0) : (dn >= 1.0);
    if (!rows_exist && far) return 0;
    if (!rows_exist && dn < 1.0) return 1;
    return 2;
}
// g = pf - (po + K h vo): what is left to the goal after coasting through the horizon (per axis)
__device__ __forceinline__ double goal_gap(double pf, double po, double vo, double h) { return pf - fma((double)K * h, vo, po); }
// a_unc(k) or w_unc(k) = 2 q g T1 + 2 s ao T2 with the two table entries of the component
__device__ __forceinline__ double unc_entry(double qw, double sw, double gax, double ao, double t1, double t2)
{
    return fma(2.0 * qw * gax, t1, (2.0 * sw * ao) * t2);
}
// A_initp(k,:) [po;vo] of component (k, axis)
__device__ __forceinline__ double init_pos(int k, double h, double vo, double po) { return fma((double)(k + 1) * h, vo, po); }
// v(k) = h sum_{kk<=k} a(kk) + vo   (propStatedmpc.m:4); a_s: the stacked accelerations in LDS
__device__ __forceinline__ double vel_out(const double *a_s, int k_l, int ax_l, double h, double vo)
{
    double sv = 0.0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) { const double ak = a_s[3 * kk + ax_l]; sv += (kk <= k_l) ? ak : 0.0; }
    return fma(h, sv, vo);
}

// The kernel's StepParams argument as it lies in the kernel-argument segment (first and only argument of every solve kernel), through a
// pointer in the CONSTANT address space: fields read through it are scalar loads (s_load, the scalar cache) issued where they are used.
// The empty asm keeps the compiler from recognising the pointer and hoisting the loads -- values kept in SGPRs across the solver loop
// were spilled to VGPR lanes.  (A generic pointer here made every field a flat VECTOR load with the pointer in VGPRs: two dependent
// memory round trips per output array in the output stage of every agent.)
typedef const StepParams __attribute__((address_space(4))) *KargPtr;
__device__ __forceinline__ KargPtr kernarg_params()
{
    KargPtr p = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Retry-ladder certificate (bounded-slack variants).  The rows of one horizon step k constrain only w_k = (Lambda a)_k,
// and with |a| <= alim the reachable set of w_k is EXACTLY the box |w_k| <= alim ((k+1) h)^2 / 2 per axis (cut by the
// workspace bounds of that step).  With every slack at its lower bound slb*f the rows are the half-spaces
// -xi.w <= b - sd slb f; if their intersection with the box is empty the QP is infeasible at that ladder level.  A
// bounded non-empty polytope has an edge on the intersection line of two of its planes, so it is empty iff on every
// such line the interval cut out by the other planes is empty: one lane per plane pair (<= 26 rows + 6 faces), one
// pass over the planes each.  Conservative (row subset, 1e-7 relative margin): a level it does not reject is still
// decided by the active-set iteration.  For solveSoftDMPCbound/bound2 all rows sit on ONE step, so the test is
// nearly exact and the ladder jumps straight to the first level that can work instead of proving 2-3 levels
// infeasible with 100+ iterations each.  Only the first 128 rows are used.  Out of line: it runs for a handful
// of agents per launch and must not cost the solver's hot loop any registers.
// `pl_planes`: planes (4 doubles each) the scratch `pl` holds -- the rows of a step beyond pl_planes - 6 are left out.  Until round 5 that was
// 26 rows everywhere (the two 64-double staging vectors), and the heaviest agents of the 10^4-agent scene are exactly those with 27-40 rows on
// their step: the certificate passed levels that the rows it did not look at make infeasible, and the dual method then proved each of them
// with 40-50 iterations at 4-5 us (tools/gpu_c4_iter_trace.py: 81 + 73 of an agent's 207 iterations).  The callers now hand over what is dead
// at the time: the three staging vectors inside a solve (44 planes), the inverse factor's block between two solves (128 rows).
__device__ __attribute__((noinline)) bool ladder_level_infeasible(const double *__restrict__ r_xi, const double *__restrict__ r_b,
                                                                  const double *__restrict__ r_sd, const double *__restrict__ r_slb,
                                                                  const int *__restrict__ r_kc, int nr, double *pl, double h,
                                                                  double alim, double f, double whi_l, double wlo_l, int lane, const int pl_planes = 32)
{
    const int mcap = pl_planes - 6;
    const int kc0 = lane < nr ? r_kc[lane] : -1, kc1 = lane + 64 < nr ? r_kc[lane + 64] : -1;
    bool empty_any = false;
    for (int k = 0; k < K && !empty_any; ++k) {
        const bool t0 = (lane < nr) && kc0 == k, t1 = (lane + 64 < nr) && kc1 == k;
        if (!__any(t0 || t1)) continue;
        const double sh = (double)(k + 1) * h, R = 0.5 * alim * sh * sh;
        // Round 5: rows that the WHOLE reachable box satisfies at this level -- max over |w| <= R of -xi.w = |xi|_1 R <= rhs -- cannot cut the
        // polytope and are left out (exact: the box faces are planes of the set).  The search is cubic in the planes -- every pair's line against
        // every plane, all of them when the level IS infeasible -- and most rows of a step belong to neighbours that only come near: at N = 10^4 a
        // ladder climber spent more time here than iterating.  A row that the whole box violates settles the level at once.
        double rx0[2], rx1[2], rx2[2], rrhs[2];
        bool keep[2], dead = false;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool sel = c ? t1 : t0;
            const int i = sel ? lane + 64 * c : 0;
            rx0[c] = r_xi[3 * i]; rx1[c] = r_xi[3 * i + 1]; rx2[c] = r_xi[3 * i + 2];
            rrhs[c] = r_b[i] - r_sd[i] * r_slb[i] * f;
            const double n1 = (fabs(rx0[c]) + fabs(rx1[c]) + fabs(rx2[c])) * R;
            keep[c] = sel && !(n1 * (1.0 + 1e-9) + 1e-12 * fabs(rrhs[c]) <= rrhs[c]);
            dead = dead || (sel && -n1 > rrhs[c] + 1e-7 * (n1 + fabs(rrhs[c])));
        }
        if (__any(dead)) { empty_any = true; break; }
        const bool s0 = keep[0], s1 = keep[1];
        const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
        const int c0 = __popcll(m0), m = c0 + __popcll(m1);
        if (m == 0) continue;
        const int M = m < mcap ? m : mcap;
        LSYNC();
        {
            const int p0 = lanes_below(m0, lane), p1 = c0 + lanes_below(m1, lane);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool sel = c ? s1 : s0;
                const int pp = c ? p1 : p0;
                if (sel && pp < mcap) {
                    const double x0 = rx0[c], x1 = rx1[c], x2 = rx2[c];
                    const double rhs = rrhs[c];
                    const double sn = 1.0 / ((fabs(x0) + fabs(x1) + fabs(x2)) * R + fabs(rhs) + 1e-300);
                    pl[4 * pp] = -x0 * sn; pl[4 * pp + 1] = -x1 * sn; pl[4 * pp + 2] = -x2 * sn; pl[4 * pp + 3] = rhs * sn;
                }
            }
            // box faces of step k: w <= min(R, whi), -w <= min(R, -wlo); the bounds of component (k, ax) live in lane 3k + ax
            if (lane >= 3 * k && lane < 3 * k + 3) {
                const int ax = lane - 3 * k;
                const double hi = fmin(R, whi_l), lo = fmax(-R, wlo_l), sn = 0.5 / R;
                double *fc = pl + 4 * (M + 2 * ax);
                fc[0] = ax == 0 ? sn : 0.0; fc[1] = ax == 1 ? sn : 0.0; fc[2] = ax == 2 ? sn : 0.0; fc[3] = hi * sn;
                fc[4] = -fc[0]; fc[5] = -fc[1]; fc[6] = -fc[2]; fc[7] = -lo * sn;
            }
        }
        LSYNC();
        const int Mt = M + 6, npair = Mt * (Mt - 1) / 2;
        double best = INFINITY;     // smallest emptiness gap over the lines seen by this lane
        bool found = false;
        for (int base = 0; base < npair && !found; base += 64) {
            const int pr = base + lane;
            const bool valid = pr < npair;
            const int n = Mt, prc = valid ? pr : 0;
            int ii = (int)((2.0f * n - 1.0f - sqrtf((2.0f * n - 1.0f) * (2.0f * n - 1.0f) - 8.0f * (float)prc)) * 0.5f);
            ii = ii < 0 ? 0 : (ii > n - 2 ? n - 2 : ii);
            while (ii > 0 && ii * (2 * n - ii - 1) / 2 > prc) --ii;
            while (ii < n - 2 && (ii + 1) * (2 * n - ii - 2) / 2 <= prc) ++ii;
            int jj = prc - ii * (2 * n - ii - 1) / 2 + ii + 1;
            jj = jj > n - 1 ? n - 1 : jj;
            const double a0 = pl[4 * ii], a1 = pl[4 * ii + 1], a2 = pl[4 * ii + 2], ab = pl[4 * ii + 3];
            const double b0 = pl[4 * jj], b1 = pl[4 * jj + 1], b2 = pl[4 * jj + 2], bb = pl[4 * jj + 3];
            // line: direction d = a x b (normalised), point w0 = (ab (b x d) + bb (d x a)) / |a x b|^2
            double d0 = a1 * b2 - a2 * b1, d1 = a2 * b0 - a0 * b2, d2 = a0 * b1 - a1 * b0;
            const double dd = d0 * d0 + d1 * d1 + d2 * d2;
            const double na2 = a0 * a0 + a1 * a1 + a2 * a2, nb2 = b0 * b0 + b1 * b1 + b2 * b2;
            const bool ok = valid && dd > 1e-16 * na2 * nb2;      // not (nearly) parallel planes
            const double idd = ok ? 1.0 / dd : 0.0;
            const double w0 = (ab * (b1 * d2 - b2 * d1) + bb * (d1 * a2 - d2 * a1)) * idd,
                         w1 = (ab * (b2 * d0 - b0 * d2) + bb * (d2 * a0 - d0 * a2)) * idd,
                         w2 = (ab * (b0 * d1 - b1 * d0) + bb * (d0 * a1 - d1 * a0)) * idd;
            const double idn = ok ? rsqrt(dd) : 0.0;
            d0 *= idn; d1 *= idn; d2 *= idn;
            // the interval [lo, hi] the other planes leave on the line, its ends kept as fractions with positive denominators
            // (lo = ln / ld, hi = hn / hd; -inf and +inf are -1/0 and 1/0): candidates are compared by cross-multiplication, two
            // divisions per line instead of one per plane (the certificate was 10-20 % of a long ladder solve)
            double ln = -1.0, ld = 0.0, hn = 1.0, hd = 0.0, par = -INFINITY;   // par: worst violation among planes parallel to the line
            const double2 *pl2 = (const double2 *)__builtin_assume_aligned(pl, 16);
            for (int c = 0; c < Mt; ++c) {
                const double2 pa = pl2[2 * c], pb = pl2[2 * c + 1];
                const double c0_ = pa.x, c1 = pa.y, c2 = pb.x;
                const double g = c0_ * d0 + c1 * d1 + c2 * d2;                        // slope along the line
                const double r = pb.y - (c0_ * w0 + c1 * w1 + c2 * w2);               // slack at w0 (relative units)
                if (g > 1e-12) { if (r * hd < hn * g) { hn = r; hd = g; } }           // hi = min(hi, r / g)
                else if (g < -1e-12) { if (-r * ld > ln * -g) { ln = -r; ld = -g; } } // lo = max(lo, r / g)
                else par = fmax(par, -r);
            }
            const double gap = fmax((ln / ld - hn / hd) / R, par);   // positive = the line misses the polytope
            if (ok) best = fmin(best, gap);
            found = __any(ok && gap <= 1e-9);
        }
        if (!found && wave_min(best) > 1e-7) empty_any = true;
    }
LSYNC();
return empty_any;
}

// --------------------------------------------------------------------------------------------
// the MPC step: a scan launch (a5/a6) and a solve launch (a7-a10, dmpc_solve.hip)
// --------------------------------------------------------------------------------------------
// scan + collision rows (a5/a6) -> global row scratch + 8-int header per agent.  Splitting the step keeps the solver's
// register and LDS footprint free of the scan's needs (more resident agents per CU) and lets the solver be re-launched
// for the few agents that overflow the tier-1 working-set capacity.
// `vb`: the agent's index in the launch (already renumbered XCD-aware by the kernel); `smem`: this wave's LDS.
// SCP (solveDMPC.m, dmpc_scp_kernel): `scp_prev` = prev_p of the pass (the previous pass's prediction of this agent, 45 doubles; null in the
// first pass: the table's own column, solveDMPC.m:10), `scp_mask` = addConstr as a bit per horizon step (in / out).
template <bool SOFT, typename TT, bool FAST, bool ORD4 = false, bool SCP = false>
__device__ __forceinline__ void scan_body(const StepParams &P, const int lane, const int vb, unsigned char *smem,
                                          const double *scp_prev = nullptr, unsigned *scp_mask = nullptr)
{
    const int S = P.S, G = P.G, C = P.C, nrmax = P.nrmax;
    const int scene = vb / P.c_count, ci = vb - scene * P.c_count;
    const int cl = P.c_first + ci;                                     // agent inside chunk g_local
    const int gid = scene * P.c_count + ci;                            // index into the launch's arrays
    const int var = P.variant;
    constexpr bool soft = SOFT;   // slack-carrying variants (bound, bound2, all3, softall, repair) vs hard rows (hard, ondemand, ellip)

    // `real`: the arithmetic (and table) type of the scan and the row builder -- double, or float in the mixed-precision mode
    // (fp32 table, distance tests and rows; the QP itself stays fp64: rows are stored as doubles)
    using real = TT;
    const TT *const tab = (const TT *)P.lT;
    const real rmin = (real)P.rmin, h_ = (real)P.h, e1z = (real)P.e1z, e2z = (real)P.e2z, alim_ = (real)P.alim;
    // safety margin of the exact pruning / certificates: far below any geometric quantity in fp64, a few float roundings of a
    // workspace-sized coordinate in the mixed mode
    constexpr real MARG = std::is_same<TT, float>::value ? (real)2e-5 : (real)1e-9;
    Lds L;
    real *own_s, *w_s;
    int *scan_cand = nullptr;
    const int *scan_nbr = nullptr;
    {
        double *p = (double *)smem;
        own_s = (real *)p; p += 48;
        w_s = (real *)p; p += 48;   // unconstrained minimiser in position space (launch-order key of the slack-free variants)
        scan_cand = (int *)p;   // SCAN_CAND_CAP ints
        // Collision rows live in a per-agent slice of a GLOBAL scratch buffer (L2-resident; lane = row, so
        // every access is a coalesced wave load): keeping them out of LDS is what lets several times more
        // agents be resident per CU.  Only the per-row working-set flags and the slack values stay in LDS.
        const size_t per = (size_t)nrmax * (soft ? 7 : 4);
        double *g = P.rowbuf + (size_t)gid * per;
        L.r_xi = g; g += 3 * nrmax;
        L.r_b = g; g += nrmax;
        if (soft) { L.r_sd = g; g += nrmax; L.r_st = g; g += nrmax; L.r_slb = g; g += nrmax; }
        else { L.r_sd = L.r_st = L.r_slb = nullptr; }
        L.r_kc = P.rowkc + (size_t)gid * nrmax;
    }
    int *hdr = P.hdr + (size_t)gid * 8;
    if (P.scene_done && __builtin_amdgcn_readfirstlane(P.scene_done[scene])) {
        // the scene's transition is over (dmpc_transition): nothing to solve, the state stays frozen (status 0 = no update)
        if (lane == 0) { hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; hdr[3] = 0; hdr[4] = 8; hdr[5] = 0; hdr[6] = 0; hdr[7] = 0; P.status[gid] = 0; }
        return;
    }

    // ---------------------------------------------------------------- agent state (uniform)
    Agent A;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        A.po[d] = P.x_p[3 * gid + d]; A.vo[d] = P.x_v[3 * gid + d];
        A.ao[d] = P.x_a[3 * gid + d]; A.pf[d] = P.pf[3 * gid + d];
    }
    // own previous prediction: prev_p = l(:,:,n)  (solveSoftDMPCbound.m:6)
    const TT *lT_own = tab + ((size_t)(P.g_local * S + scene) * N3) * C + cl;
    if (lane < N3) own_s[lane] = (SCP && scp_prev) ? (real)scp_prev[lane] : lT_own[(size_t)lane * C];
    LSYNC();
    real po_[3], vo_[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { po_[d] = (real)A.po[d]; vo_[d] = (real)A.vo[d]; }

    // ---------------------------------------------------------------- a5/a6: scan + rows
    int nr = 0, nrows_ref = 0, viol_k = 0, status = 0;
    bool cert_infeasible = false;
    int ladder_start = 0;   // first retry-ladder level that is not certainly infeasible (soft ladder variants)
    bool rows_exist = false, violation = false;
    const bool cppv = (var == VAR_CPP || var == VAR_CPP2);   // dmpc/cpp solveQPv2 flavour (dmpc.cpp:803-1287)
    // Super-ellipsoid of ORDER 4 (round 4; CheckCollEllipDMPC.m:7, CollConstrEllipDMPC.m:13-19 with order = 4, E1 = E^-1, E2 = E^-4 as
    // test/comp_test_ellipconstr.m:158-187 sets them for solveSoftDMPC): dist = |E1 d|_4, xi = E2 d.^3, prev_dist = dist^3,
    // r = dist^3 (rmin - dist) + xi.p - xi.A0 x0.  Only the variants whose scan is "any neighbour inside rmin, rows for every neighbour"
    // (softall, ellip, repair, cpp1: the generic walk below + build_rows) take it; the API refuses the others.
    constexpr bool ord4 = ORD4;   // (a template parameter: the order-2 kernels carry none of it)
    const bool near_sel = (var == VAR_BOUND || var == VAR_BOUND2 || var == VAR_ALL3 || var == VAR_ONDEMAND || cppv);
    const bool coll_check = (var == VAR_BOUND || var == VAR_BOUND2 || var == VAR_ALL3 || var == VAR_REPAIR);
    const bool skip_k1 = (var == VAR_BOUND2 || var == VAR_ALL3 || var == VAR_REPAIR || var == VAR_CPP2);
    bool coll_flag = false;
    // Neighbour list (large scenes, nbr_kernel): a neighbour can come within ellipsoidal distance R of the agent at some
    // horizon step only if the bounding boxes of the two predicted horizons are within R per axis.  The survivors of
    // that test (3 % of the scene at N = 10^4) come, in increasing index order, from the pre-pass; the distance tests /
    // row builders below walk the list instead of all N neighbours.  Conservative: results are unchanged.
    // nnbr < 0: no list (small scenes, variants that take every neighbour, or more survivors than the list holds).
    int nnbr = -1;
    if (P.nbr_cnt) {
        // the pre-pass leaves the list in NBR_PARTS pieces (one per quarter of the scene, each in increasing neighbour order): close the gaps
        int *lst = P.nbr_list + (size_t)gid * P.nbr_cap;
        const int pcap = P.nbr_cap / NBR_PARTS;
        int tot = 0;
        bool fits = true;
        for (int q = 0; q < NBR_PARTS; ++q) {
            const int n = __builtin_amdgcn_readfirstlane(P.nbr_cnt[(size_t)gid * NBR_PARTS + q]);   // (a vector load of a wave-uniform value: the list length steers every loop of the walk)
            if (n < 0) { fits = false; break; }
            if (q > 0 && tot < q * pcap)
                for (int i = lane; i < n; i += 64) lst[tot + i] = lst[q * pcap + i];   // moves down: a round's reads are at or above its writes
            tot += n;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (fits) { nnbr = tot; scan_nbr = lst; }
    }
    // the neighbours are walked 64 at a time: entry e0 + lane of the list, or neighbour (r, j0 + lane) of the table
    const int cpad = (C + 63) & ~63;
    const int n_entries = nnbr >= 0 ? nnbr : G * cpad;
    auto entry = [&](int e0, int &r, int &jc) -> bool {
        if (nnbr >= 0) {
            const bool have = e0 + lane < nnbr;
            const int code = have ? scan_nbr[e0 + lane] : 0;
            r = code >> 20; jc = code & 0xfffff;
            return have;
        }
        r = e0 / cpad;
        const int jj = e0 - r * cpad + lane;
        jc = jj < C ? jj : C - 1;   // clamped: loads are unconditional, results masked
        // unequal clusters (dmpc.cpp:1600-1625: the first N mod G clusters hold one agent more): the chunks from
        // P.short_from on use C-1 columns, their last column is padding
        return jj < C - ((P.short_from && r >= P.short_from) ? 1 : 0) && !(r == P.g_local && jj == cl);
    };

    // Launch-order key of the slack-free variants (see order_kernel): which horizon steps have a row that is violated
    // at the unconstrained minimiser, and how tight the tightest row is against the reachable box.
    unsigned key_steps = 0, key_tight = 0;
    float key_soft = 0.f;    // slack variants: the launch-order key before the terms of the unconstrained-exit block
    float key_share = 1.f;   // slack ladder variants: the smallest share of the reachable box a row leaves feasible at ladder level 0
    if (!soft) {
        // w_unc = Lambda a_unc of the collision cost case (rows exist): 2 q g P1[k][K-1] + 2 s ao M1[0][k]  (Gram table
        // of case 2: G[W k][W K-1] and G[W k][A 0])
        if (lane < N3) {
            const int k = lane / 3, ax = lane - 3 * k;
            const double *tb = P.tables + 2 * TAB_CASE_DOUBLES;
            const double gax = sel3(A.pf, ax) - (sel3(A.po, ax) + (double)K * h_ * sel3(A.vo, ax));
            w_s[lane] = (real)(2.0 * P.Q1 * gax * tb[(15 + k) * 30 + 15 + (K - 1)] + 2.0 * P.S1 * sel3(A.ao, ax) * tb[(15 + k) * 30]);
        }
        LSYNC();
    }
    {
        // appends the rows of horizon step ke (evaluated positions) constraining step kc for every
        // neighbour with dist < sel_r (or all), in increasing neighbour index (CollConstrSoftDMPC.m:11-31)
        // one collision row, CollConstrSoftDMPC.m:16-28: neighbour offset (dx,dy,dz) and ellipsoidal distance at
        // the evaluation step (own position px,py,pz), constraining horizon step kc; compacted by ballot
        auto emit_row = [&](bool sel, int kc, real dx, real dy, real dz, real dist, real px, real py, real pz) {
            const real sh = (real)(kc + 1) * h_;
            const real a0x = po_[0] + sh * vo_[0], a0y = po_[1] + sh * vo_[1], a0z = po_[2] + sh * vo_[2];
            // diff = E2*(p - pj).^(order-1); pd = prev_dist = dist^(order-1) (P.e2z = c^-order)
            // (order 4, DMPC::solveQP: dmpc.cpp:47,478 scale before the power, (E2 d).^3 -- z component (c^-4 dz)^3; the MATLAB helpers E2 * d.^3)
            const real dzs = dz * e2z;
            const real x0 = ord4 ? dx * dx * dx : dx, x1 = ord4 ? dy * dy * dy : dy, x2 = ord4 ? (var == VAR_CPP1 ? dzs * dzs * dzs : dz * dz * dz * e2z) : dzs;
            const real pd = ord4 ? dist * dist * dist : dist;
            {
                const unsigned long long m0 = __ballot(sel);
                if (m0 == 0ull) return;                         // no neighbour of this chunk is close at this step
                nrows_ref += __popcll(m0);                      // the reference's row count (branch record)
            }
            // Exact pruning: with |a| <= alim the position at step kc stays in the box
            // A0_kc x0 +- alim (kc h)^2/2, so a row whose linearised distance cannot drop below rmin anywhere in
            // that box can never become active; dropping it (and its slack, which stays 0) leaves the minimiser
            // unchanged.  Margin 1e-9 keeps borderline rows.
            {
                const real hw = 0.5 * alim_ * sh * sh;
                if (!soft && sel) {
                    // the row reads  xi.w_kc >= rr  for the position offset w = Lambda a;  |w_kc| <= hw per axis
                    // (the right-hand side of the row without the reference's division: dist (xi.p / dist) = xi.p)
                    const real rr = pd * (rmin - dist) + (x0 * px + x1 * py + x2 * pz) - (x0 * a0x + x1 * a0y + x2 * a0z);
                    const real rng = (fabs(x0) + fabs(x1) + fabs(x2)) * hw;
                    if (rr - (x0 * w_s[3 * kc] + x1 * w_s[3 * kc + 1] + x2 * w_s[3 * kc + 2]) > (real)0.1 * MARG) key_steps |= 1u << kc;
                    key_tight |= (rr > 0.8 * rng) ? 8u : ((rr > 0.5 * rng) ? 4u : ((rr > 0.0) ? 2u : 1u));
                }
                const real lin_min = x0 * (a0x - (px - dx)) + x1 * (a0y - (py - dy)) + x2 * (a0z - (pz - dz)) - (fabs(x0) + fabs(x1) + fabs(x2)) * hw;
                if (lin_min >= pd * rmin + MARG && !P.no_prune) sel = false;
                // Exact infeasibility certificate for rows without slack: if even the BEST point of the
                // reachable box violates the row (max of the linearised distance < dist*rmin), no acceleration
                // within |a| <= alim satisfies it -> the QP is infeasible; the long active-set proof is skipped.
                if (sel) {
                    const real lin_max = lin_min + 2.0 * (fabs(x0) + fabs(x1) + fabs(x2)) * hw;
                    if (!soft) {
                        if (lin_max < pd * rmin - MARG) cert_infeasible = true;
                    } else if (var == VAR_BOUND || var == VAR_BOUND2 || var == VAR_ALL3 || cppv) {
                        // Soft rows with a bounded slack (coefficient dist): the row needs
                        //   lin >= dist*(rmin + eps),  eps >= slb * 2^t at ladder level t (solveSoftDMPCbound.m:147-153),
                        // so every level with lin_max < dist*(rmin + slb 2^t) is certainly infeasible and the
                        // retry ladder can start at the first level that passes this necessary test.
                        real slb_t = (var == VAR_BOUND) ? -0.05 : (cppv ? -(real)0.01f : -0.01);
                        {   // launch-order key: the feasible share of the reachable box at ladder level 0 (as the slack-free variants' tightness class)
                            const real share = (lin_max - dist * (rmin + slb_t)) / (lin_max - lin_min + (real)1e-30);
                            key_share = fminf(key_share, (float)share);
                        }
                        int t = 0;
                        while (t < 40 && lin_max < dist * (rmin + slb_t) - MARG) { slb_t *= 2.0; ++t; }
                        if (t > ladder_start) ladder_start = t;
                    }
                }
            }
            const unsigned long long m = __ballot(sel);
            const int pos = nr + lanes_below(m, lane);
            if (sel && pos < nrmax) {
                // r = dist*(rmin - dist + diff*p/dist) - diff*A_initp(kc)*[po;vo]   (:21)
                const real rr = pd * (rmin - dist + (x0 * px + x1 * py + x2 * pz) / pd) - (x0 * a0x + x1 * a0y + x2 * a0z);
                L.r_xi[3 * pos] = x0; L.r_xi[3 * pos + 1] = x1; L.r_xi[3 * pos + 2] = x2;
                L.r_b[pos] = -rr;
                L.r_kc[pos] = kc;
                    if (soft) {
                    real sd = pd, st = P.term, slb = -0.05;
                    if (var == VAR_BOUND2 || var == VAR_ALL3) slb = -0.01;           // bound2:77, all:92
                    else if (cppv) slb = -(real)0.01f;                              // dmpc.cpp:907-914,1079: -eps <= lim, float lim = 0.01
                    else if (var == VAR_SOFTALL) { sd = 1.0; st = -1e5; slb = -INFINITY; }  // solveSoftDMPC.m:21,65
                    else if (var == VAR_SOFTALL_C) {
                        // solveSoftDMPC_c.m:18-20,60-63: rows [Ainr I], cost EPS eps^2 + f_eps eps with EPS = 1e6 (K/k)^2, f_eps = -1e4 (K/k)^2, k the
                        // violating step (1-based).  The solver's slack variables carry the unit weight of every other variant (H_eps = 2): substitute
                        // eps = eps' / sqrt(EPS) -- the row's slack coefficient becomes 1 / sqrt(EPS), the linear cost f_eps / sqrt(EPS), the pin
                        // eps <= 0 is eps' <= 0; the minimiser in a is the same.
                        const double rk = (double)K / (double)(kc + 1), quad = 1.0 * 1e6 * (rk * rk), lin = -1.0 * 1e4 * (rk * rk);
                        const double isq = 1.0 / sqrt(quad);
                        sd = (real)isq; st = (real)(lin * isq); slb = -INFINITY;
                    }
                    else if (var == VAR_CPP1) { sd = 1.0; st = -1e6; slb = -INFINITY; }     // dmpc.cpp:629-633,715: [A I] x <= b, eps <= 0, f_w = -10^6
                    else if (var == VAR_REPAIR) { st = P.term / pd; slb = -INFINITY; }      // repair:77,81 (term ./ prev_dist)
                    L.r_sd[pos] = sd; L.r_st[pos] = st; L.r_slb[pos] = slb;
                }
            }
            nr += __popcll(m);
        };

        // rows of horizon step ke (positions evaluated there) constraining step kc for every neighbour whose
        // distance AT STEP ksel is < sel_r (viol_constr of CheckCollSoftDMPC.m:12) or for all neighbours,
        // in increasing neighbour index (CollConstrSoftDMPC.m:11-31)
        auto build_rows = [&](int ksel, int ke, int kc, real sel_r, bool sel_all) {
            const real qx = own_s[3 * ksel], qy = own_s[3 * ksel + 1], qz = own_s[3 * ksel + 2];
            const real px = own_s[3 * ke], py = own_s[3 * ke + 1], pz = own_s[3 * ke + 2];
            for (int e0 = 0; e0 < n_entries; e0 += 64) {
                int r, jc;
                const bool valid = entry(e0, r, jc);
                const real *nbp = tab + ((size_t)(r * S + scene) * N3) * C + jc;
                const real *base = nbp + (size_t)(3 * ke) * C, *bsel = nbp + (size_t)(3 * ksel) * C;
                const real dx = px - base[0], dy = py - base[(size_t)C], dz = pz - base[2 * (size_t)C];
                const real ez = dz * e1z;
                const real dist = ord4 ? sqrt(sqrt(dx * dx * dx * dx + dy * dy * dy * dy + ez * ez * ez * ez)) : sqrt(dx * dx + dy * dy + ez * ez);
                real dsel = dist;
                if (ksel != ke && !sel_all) {
                    const real sx = qx - bsel[0], sy = qy - bsel[(size_t)C];
                    const real sz = (qz - bsel[2 * (size_t)C]) * e1z;
                    dsel = sqrt(sx * sx + sy * sy + sz * sz);
                }
                emit_row(valid && (sel_all || dsel < sel_r), kc, dx, dy, dz, dist, px, py, pz);
            }
        };

        // Scan (CheckCollSoftDMPC.m:7-15): all K distances of a neighbour are computed from loads issued
        // together (5 horizon steps = 15 coalesced wave loads per batch) instead of one dependent round per
        // step; per-step "any neighbour inside rmin" bits are OR-reduced across the wave afterwards.
        int ncand = 0;
        // second pass of the hard-row scan: one lane per buffered candidate
        auto flush_candidates = [&]() {
            LSYNC();
            for (int b0 = 0; b0 < ncand; b0 += 64) {
                const bool have = b0 + lane < ncand;
                const int code = have ? scan_cand[b0 + lane] : 0;
                const int kk = (code >> 28) & 15, rr = (code >> 20) & 255, jj = code & 0xfffff;
                const real *nb = tab + ((size_t)(rr * S + scene) * N3 + 3 * kk) * C + jj;
                const real px = own_s[3 * kk], py = own_s[3 * kk + 1], pz = own_s[3 * kk + 2];
                const real dx = px - nb[0], dy = py - nb[(size_t)C], dz = pz - nb[2 * (size_t)C];
                const real ez = dz * e1z;
                const real dist = sqrt(dx * dx + dy * dy + ez * ez);
                emit_row(have && dist < 1.0, kk, dx, dy, dz, dist, px, py, pz);
            }
            ncand = 0;
            LSYNC();
        };
        unsigned anyb = 0;
        real mind0 = INFINITY;
        const real rmin2_hi = (ord4 ? rmin * rmin * rmin * rmin : rmin * rmin) * ((real)1.0 + (real)4.0 * MARG);
        if (var == VAR_HARD) {
            // solveHardDMPC.m:18-22 + CollConstrHardDMPC.m:19: every k, neighbours with dist < 1.  Two passes.  The first
            // only tests the 15 N distances and compacts the (step, neighbour) candidates -- about one in ten -- into an
            // LDS list; flush_candidates() then builds the rows of 64 candidates at a time, so the row arithmetic runs on
            // full waves.  The first pass walks the FLAT index e = k * ne + entry (step-major, the reference's row order),
            // 64 pairs per round: every round is a full wave whatever N is (lanes = neighbours wastes 22 % of the issue
            // slots at N = 100, and the scan is instruction-issue bound), a lane's three loads share one address
            // computation, and runs of lanes with the same k still read contiguous table rows.
            const unsigned ne = (unsigned)(nnbr >= 0 ? nnbr : G * C);
            const unsigned total = ne * (unsigned)K;
            // (k, entry) of a lane advance by 64 pairs per round: running counters with one wrap test, no division in the
            // loop (64 = step_q * ne + step_r; lanes past the end have k >= K)
            const UDiv div_ne(ne), div_c((unsigned)C);
            const int step_q = (int)(64u / (ne ? ne : 1u)), step_r = (int)(64u - (unsigned)step_q * ne);
            const size_t slab = (size_t)N3 * C;
            auto pass = [&](auto single_tag) {
                constexpr bool single = decltype(single_tag)::value;   // one chunk, no neighbour list: entry == neighbour index
                constexpr int UR = 4;   // rounds in flight: the loads of four rounds are issued before the first is used
                int k_run, idx_run;
                div_ne.divmod((unsigned)lane, k_run, idx_run);
                for (unsigned e0 = 0; e0 < total; e0 += 64 * UR) {
                    // room for a whole group of rounds is made BEFORE its loads are issued, so that the row builder never
                    // runs with the group's registers live
                    if (ncand + 64 * UR > SCAN_CAND_CAP) flush_candidates();
                    real nx[UR], ny[UR], nz[UR];
                    int kk[UR], code[UR];
                    bool ok[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        bool valid = k_run < K;
                        const int k = valid ? k_run : 0, idx = valid ? idx_run : 0;
                        idx_run += step_r; k_run += step_q;
                        if (idx_run >= (int)ne) { idx_run -= (int)ne; k_run += 1; }
                        int r = 0, jc = idx;
                        if (!single) {
                            if (nnbr >= 0) {
                                const int c = scan_nbr[idx];
                                r = c >> 20; jc = c & 0xfffff;
                            } else {
                                div_c.divmod((unsigned)idx, r, jc);
                                valid = valid && !(r == P.g_local && jc == cl) && jc < C - ((P.short_from && r >= P.short_from) ? 1 : 0);
                            }
                        } else valid = valid && jc != cl;
                        const real *nb = tab + ((size_t)(single ? P.g_local : r) * S + scene) * slab + (unsigned)(3 * k * C + jc);
                        nx[u] = nb[0]; ny[u] = nb[(size_t)C]; nz[u] = nb[2 * (size_t)C];
                        kk[u] = k; code[u] = (k << 28) | (r << 20) | jc; ok[u] = valid;
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        if (e0 + 64u * u >= total) break;
                        const int k = kk[u];
                        const real px = own_s[3 * k], py = own_s[3 * k + 1], pz = own_s[3 * k + 2];
                        const real dx = px - nx[u], dy = py - ny[u], dz = pz - nz[u];
                        const real ez = dz * e1z;
                        const real d2 = dx * dx + dy * dy + ez * ez;
                        // squared distance against a slightly inflated threshold (a superset); the exact `norm(...) < 1`
                        // decision is made on the IEEE square root in the second pass
                        const bool cand = ok[u] && d2 < (real)1.0 + (real)4.0 * MARG;
                        const unsigned long long cm = __ballot(cand);
                        if (cm) {
                            if (cand) scan_cand[ncand + lanes_below(cm, lane)] = code[u];
                            ncand += __popcll(cm);
                        }
                    }
                }
            };
            if (G == 1 && nnbr < 0) pass(std::true_type{});
            else pass(std::false_type{});
        } else if (nnbr >= 0 && P.lrow) {
            // Neighbour list: the walk is TRANSPOSED -- lanes = horizon components, one listed neighbour per step of the loop.
            // With lanes = neighbours every table load of a round is a gather of 64 cache lines (45 of them per round: at
            // N = 10^4 the list walk took 1-2 ms of gathers); here a neighbour's whole horizon is ONE coalesced 256-byte load
            // from the neighbour-major fp32 copy of the table (table_nbrmajor_kernel: [chunk][scene][column][15 x (x, y, z, 0) + 4];
            // 2.5 MB at N = 10^4: L2 resident), and the three squares of a step are summed with two row-shift DPP moves (groups
            // of 4 lanes never straddle a row).  The fp32 distance only SELECTS (threshold widened by 1e-3, a hundred times the
            // rounding of a workspace-sized coordinate); the decision dist < rmin is made on the table itself, with the
            // arithmetic of the other walk, for the few (step, neighbour) pairs that pass.
            const float *rt = (const float *)P.lrow;
            const float *rt1 = rt + (size_t)scene * C * 64;
            constexpr int TW = 8;
            const int k4 = lane >> 2, a4 = lane & 3;
            const bool act = a4 < 3 && k4 < K;
            const bool head = a4 == 0 && k4 < K;
            const float ownc = act ? (float)own_s[3 * k4 + a4] : 0.f;
            const float sc = act ? (a4 == 2 ? (float)e1z : 1.f) : 0.f;
            const float thr = (float)(rmin * rmin) * 1.001f;
            const int k3 = head ? 3 * k4 : 0;
            // (two copies of the loop: one chunk -- a scene on one GPU, the code IS the column -- and the general form, which costs eight more
            // scalar instructions per listed neighbour; a test per neighbour cost branches instead.  The walk of the 100-700 neighbours of an
            // agent of a 10^4-agent scene is scalar-bound.)
            auto walk = [&](auto one_tag) {
            constexpr bool one = decltype(one_tag)::value;
            for (int e0 = 0; e0 < nnbr; e0 += 64) {
                const int codes = (e0 + lane < nnbr) ? scan_nbr[e0 + lane] : 0;
                const int m = (nnbr - e0) < 64 ? (nnbr - e0) : 64;
                for (int u0 = 0; u0 < m; u0 += TW) {
                    float nv[TW];   // loads in flight
                    int cu[TW];
#pragma unroll
                    for (int u = 0; u < TW; ++u) {
                        const int uu = (u0 + u < m) ? u0 + u : m - 1;   // (a repeated neighbour changes nothing: OR and min)
                        cu[u] = readlane_i(codes, uu);
                        nv[u] = one ? rt1[(size_t)(unsigned)cu[u] * 64 + lane]
                                    : rt[((size_t)((cu[u] >> 20) * S + scene) * C + (cu[u] & 0xfffff)) * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < TW; ++u) {
                        const float d = (ownc - nv[u]) * sc;
                        const float q2 = d * d;
                        const float d2f = (q2 + dpp_row_shl<1>(q2)) + dpp_row_shl<2>(q2);   // lanes 4k: squared distance at step k
                        const bool flag = head && d2f < thr;
                        if (__any(flag)) {
                            const real *nb = tab + ((size_t)((cu[u] >> 20) * S + scene) * N3 + k3) * C + (cu[u] & 0xfffff);
                            if (flag) {
                                const real dx = own_s[k3] - nb[0], dy = own_s[k3 + 1] - nb[(size_t)C], dz = own_s[k3 + 2] - nb[2 * (size_t)C];
                                const real ez = dz * e1z;
                                const real d2 = dx * dx + dy * dy + ez * ez;
                                const real dist = sqrt(d2);
                                if (dist < rmin) anyb |= (1u << k4);       // CheckCollSoftDMPC.m:11
                                if (k4 == 0) mind0 = fmin(mind0, d2);       // (only ever used when some neighbour is inside rmin at k = 0)
                            }
                        }
                    }
                }
            }
            };
            if (G == 1) walk(std::true_type{}); else walk(std::false_type{});
        } else {
            for (int e0 = 0; e0 < n_entries; e0 += 64) {
                int r, jc;
                const bool valid = entry(e0, r, jc);
                const real *base = tab + ((size_t)(r * S + scene) * N3) * C + jc;
                // the 45 component rows of a neighbour are C doubles apart: a running (uniform) offset, advanced by one
                // 64-bit scalar add per load (written as (3k+c)*C the compiler spends five scalar instructions per load)
                size_t roff = 0;
    #pragma unroll 1
                for (int kg = 0; kg < 3; ++kg) {   // not unrolled: keeps the scan's register footprint small
                    real nx[5], ny[5], nz[5];
    #pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        nx[u] = base[roff]; roff += (size_t)C; if (!SCP) asm volatile("" : "+s"(roff));   // (not in the SCP instantiation: inside dmpc_scp_kernel's pass loop the compiler keeps the offset in a vector register and refuses the constraint)
                        ny[u] = base[roff]; roff += (size_t)C; if (!SCP) asm volatile("" : "+s"(roff));
                        nz[u] = base[roff]; roff += (size_t)C; if (!SCP) asm volatile("" : "+s"(roff));
                    }
    #pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int k = 5 * kg + u;
                        const real px = own_s[3 * k], py = own_s[3 * k + 1], pz = own_s[3 * k + 2];
                        const real dx = px - nx[u], dy = py - ny[u], dz = pz - nz[u];
                        const real ez = dz * e1z;
                        // (order 4: the sum of fourth powers against rmin^4; the root of the root decides)
                        const real d2 = ord4 ? dx * dx * dx * dx + dy * dy * dy * dy + ez * ez * ez * ez : dx * dx + dy * dy + ez * ez;
                        // The IEEE square root (a dozen instructions in fp64) is taken only when some lane of the
                        // wave can pass the distance test: d2 is first compared against a slightly inflated squared
                        // threshold (a superset), the exact `norm(...) < r` decision is then made on sqrt(d2) itself.
                        if (__any(valid && d2 < rmin2_hi)) {
                            // (the empty asm keeps this a BRANCH: the compiler turned the wave-uniform test into selects and ran the 25
                            // instructions of the IEEE root for every step of every neighbour -- 800 of the 3 000 instructions of an agent's scan)
                            asm volatile("");
                            const real dist = ord4 ? sqrt(sqrt(d2)) : sqrt(d2);
                            if (valid && dist < rmin) anyb |= (1u << k);       // CheckCollSoftDMPC.m:11
                        }
                        if (k == 0 && valid) mind0 = fmin(mind0, d2);           // squared; the root is taken once below
                    }
                }
            }
        }
        if (SCP) {
            // solveDMPC.m:21-35: CheckCollDMPC at every step of prev_p (anyb: the walk above with E1 = I); rows for ALL other agents at every step
            // of addConstr, and at the FIRST violating step that is not in it yet (at most one new step per pass); ascending k = the reference's order
            anyb = wave_or(anyb);
            unsigned m = *scp_mask;
            bool newc = false;
            for (int k = 0; k < K; ++k) {
                bool add = (m >> k) & 1u;
                if (!add && !newc && ((anyb >> k) & 1u)) { add = true; newc = true; m |= 1u << k; }
                if (add) build_rows(k, k, k, (real)0, true);
            }
            *scp_mask = m;
            rows_exist = nrows_ref > 0;               // `isempty(Ain_total)` (:38)
            violation = m != 0u;
            viol_k = m ? __ffs((int)m) : 0;           // smallest member of addConstr, 1-based
        } else if (var == VAR_HARD) {
            flush_candidates();
            rows_exist = (G * C - (P.short_from ? G - P.short_from : 0) > 1);   // preallocated zero rows make Ain_coll non-empty (CollConstrHardDMPC.m:3-4)
        } else {
            anyb = wave_or(anyb);
            for (int k = 0; k < K; ++k) {
                if (!((anyb >> k) & 1u)) continue;
                if (var == VAR_ALL3) violation = true;   // solveSoftDMPCall.m:22 (some_violation)
                if ((coll_check || cppv) && k == 0) {
                    mind0 = sqrt((real)wave_min((double)mind0));   // min of the roots == root of the min (sqrt is monotone)
                    if (ord4) mind0 = sqrt(mind0);                // (order 4: the walk accumulated sums of fourth powers)
                    if (coll_check && mind0 < rmin - 0.05) { status = ST_COLL; viol_k = 1; break; }   // :25-31
                    // cpp: `dist < _rmin - _collision_tol` (floats) raises execution_ended, the build goes on (dmpc.cpp:419-424)
                    if (cppv && mind0 < (real)((float)rmin - 0.05f)) coll_flag = true;
                }
                // DMPC::solveQP (dmpc.cpp:626-637): rows on step k-1 (`3*(k-1)`, :480-485); k = 0 indexes row -3 there (undefined): `coll`, no QP
                if (var == VAR_CPP1 && k == 0) { status = ST_COLL; viol_k = 1; break; }
                if (skip_k1 && k == 0) continue;          // solveSoftDMPCbound2.m:29-31
                viol_k = k + 1; violation = true; rows_exist = true;
                if (var == VAR_ALL3) {                     // solveSoftDMPCall.m:34-48: steps k-1,k,k+1
                    const int k0 = (k == 1) ? k : k - 1, k1 = (k == K - 1) ? k : k + 1;
                    for (int kk = k0; kk <= k1; ++kk) build_rows(k, kk, kk, 3.0 * rmin, false);
                } else {
                    const int kc = (var == VAR_BOUND2 || var == VAR_CPP2 || var == VAR_CPP1) ? k - 1 : k;   // CollConstrSoftDMPC2.m:8; dmpc.cpp:516; :480-485
                    // neighbour radius: 3 rmin (CheckCollSoftDMPC.m:12); cpp: _rmin*(1+(float)k/_k_hor) in float arithmetic
                    // (dmpc.cpp:418) -- __f*_rn keep the three float operations unfused
                    const real near_r = cppv ? (real)__fmul_rn((float)rmin, __fadd_rn(1.0f, __fdiv_rn((float)k, (float)K))) : 3.0 * rmin;
                    build_rows(k, k, kc, near_r, !near_sel);
                }
                break;
            }
        }
        // header for the solve phase
        if (nr > nrmax) { status |= ST_CAPACITY; nr = nrmax; }
        if (__any(cert_infeasible)) status |= ST_INFEAS;
        if (lane == 0) {
            hdr[0] = nr; hdr[1] = nrows_ref; hdr[2] = viol_k; hdr[3] = status;
            hdr[4] = (violation ? 1 : 0) | (coll_flag ? 4 : 0); hdr[5] = rows_exist ? 1 : 0; hdr[6] = 0;
        }
        const int ls_key = (int)wave_max((double)ladder_start);
        if (lane == 0) hdr[6] = ls_key;
        {
            // launch-order key (order_kernel, heaviest first).  Slack-carrying variants: the row count.  Slack-free
            // variants: 4 x (horizon steps with a row violated at the unconstrained minimiser) + tightness class of the
            // tightest row (its feasible share of the reachable box: > 1/2, > 1/4, > 1/10, less).  Forcing launch orders on
            // the hardware (tools/gpu_order_probe.py, 51 200 C2 agents): natural 1240 us, by row count 1137, by the true
            // iteration counts 1096, by this key 1064.
            int key = nr >> 2;
            if (soft) {
                // Slack variants (round 4): rows / 4 said little about the work -- at N = 10^4 the solve launch took 1.6 times its work per wave
                // slot and ended with agents of 100-450 us from the END of the queue.  Ingredients fitted on that scene's closed loop
                // (tools/gpu_key_features.py + tools/key_fit.py: list scheduling of the agents' measured work in the order a candidate key gives):
                // the smallest feasible share of the reachable box a row leaves at ladder level 0, the ladder levels the scan certified
                // infeasible, and -- added in the unconstrained-exit block below, where the unconstrained minimiser is known -- the acceleration
                // bounds and the rows it violates.  (An agent's own previous solve predicts the bulk, rank correlation 0.8, but not the heavy
                // agents: a retry ladder is a one-step event.  The perfect order would be worth 0.97 -> 0.72 ms; this key 0.97 -> 0.8x.)
                const float smin = -(float)wave_max((double)-key_share);
                key_soft = 0.7f * (float)nr + 56.f * (1.f - fminf(fmaxf(smin, 0.f), 1.f)) + 52.f * (float)ls_key;
                key = (int)key_soft;
            }
            if (!soft) {
                const unsigned ks = wave_or(key_steps), kt = wave_or(key_tight);
                key = 4 * __popc(ks) + (kt ? 31 - __clz((int)kt) : 0);
            }
            if (lane == 0) hdr[7] = key > 255 ? 255 : key;
        }
        // ---- unconstrained exit.  Most agent-steps of a transition are trivial: no bound, workspace wall or collision row is violated
        // at the unconstrained minimiser a_unc (closed form from the tables), so the active-set loop would stop before its first
        // iteration.  Such an agent is finished HERE -- same arithmetic as the solver's set-up and output stage (the shared helpers
        // above), its header flagged (hdr[4] & 16) -- and never enters the solve queue: the scan runs at twice the solver's
        // occupancy and the agent's state is already in registers.  (43 % of the agents of the solveSoftDMPCbound replay of
        // bench.py, nearly all of them once the swarm has spread out.)
        if (FAST && status == 0 && !__any(cert_infeasible) && nr <= 128) {   // (FAST: a template parameter -- launches without the exit do not carry its registers)
            const int ls = (int)wave_max((double)ladder_start);
            // The launch parameters and the agent's state are read AGAIN here -- scalar loads through constant-address-space pointers --
            // instead of staying live in SGPRs across the neighbour walk and the row builder: the slack variants' scan spilled 190
            // scalar registers to VGPR lanes (a fifth of its instructions were v_readlane / v_writelane and the s_nop around them).
            const KargPtr Qp = kernarg_params();
            typedef const double __attribute__((address_space(4))) *ConstD;
            Agent B_;
            {
                const ConstD sp = (ConstD)(unsigned long long)(Qp->x_p + 3 * (size_t)gid), sv = (ConstD)(unsigned long long)(Qp->x_v + 3 * (size_t)gid);
                const ConstD sa = (ConstD)(unsigned long long)(Qp->x_a + 3 * (size_t)gid), sf = (ConstD)(unsigned long long)(Qp->pf + 3 * (size_t)gid);
#pragma unroll
                for (int d = 0; d < 3; ++d) { B_.po[d] = sp[d]; B_.vo[d] = sv[d]; B_.ao[d] = sa[d]; B_.pf[d] = sf[d]; }
            }
            const int ccase = cost_case(var, B_.po[0] - B_.pf[0], B_.po[1] - B_.pf[1], B_.po[2] - B_.pf[2], rows_exist);
            const double qw = ccase == 0 ? Qp->Qfar : (ccase == 1 ? Qp->Qnear : Qp->Q1);
            const double sw = ccase == 2 ? ((var == VAR_ALL3) ? 10.0 : Qp->S1) : Qp->Sfree;
            const bool comp = lane < N3;
            const int k_l = comp ? lane / 3 : 0, ax_l = comp ? lane - 3 * k_l : 0;
            const double *tb = Qp->tables + (size_t)ccase * TAB_CASE_DOUBLES;
            double a_unc = 0.0, w_unc = 0.0, p0_l = 0.0, vo_l = 0.0;
            bool viol = false;
            const double tol = 1e-10;
            if (comp) {
                const double gax = goal_gap(sel3(B_.pf, ax_l), sel3(B_.po, ax_l), sel3(B_.vo, ax_l), Qp->h);
                const double ao_l = sel3(B_.ao, ax_l);
                a_unc = unc_entry(qw, sw, gax, ao_l, tb[k_l * 30 + 15 + (K - 1)], tb[k_l * 30]);
                w_unc = unc_entry(qw, sw, gax, ao_l, tb[(15 + k_l) * 30 + 15 + (K - 1)], tb[(15 + k_l) * 30]);
                vo_l = sel3(B_.vo, ax_l);
                const double sh = (double)(k_l + 1) * Qp->h * vo_l;
                const double whi = (ax_l == 0 ? Qp->pmax[0] : (ax_l == 1 ? Qp->pmax[1] : Qp->pmax[2])) - sel3(B_.po, ax_l) - sh, wlo = (ax_l == 0 ? Qp->pmin[0] : (ax_l == 1 ? Qp->pmin[1] : Qp->pmin[2])) - sel3(B_.po, ax_l) - sh;
                p0_l = init_pos(k_l, Qp->h, vo_l, sel3(B_.po, ax_l));
                viol = (fabs(a_unc) - Qp->alim > tol) || (fmax(w_unc - whi, wlo - w_unc) > tol);
            }
            bool trivial = ls == 0 && !__any(viol);
            double *a_s = (double *)smem, *wu_s = a_s + 48;   // (own prediction and key vector are dead by now)
            if (soft) {   // launch-order key, second part: acceleration bounds and rows violated at the unconstrained minimiser
                const int nsat = __popcll(__ballot(comp && fabs(a_unc) - Qp->alim > tol));
                LSYNC();
                if (comp) { a_s[lane] = a_unc; wu_s[lane] = w_unc; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the rows this wave wrote are read back
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                LSYNC();
                int nrv = 0;   // (wave-uniform: whole rounds)
                for (int i0 = 0; i0 < nr; i0 += 64) {
                    const bool in = i0 + lane < nr;
                    const int i = in ? i0 + lane : 0;
                    const int kc = L.r_kc[i];
                    const double v = -(L.r_xi[3 * i] * wu_s[3 * kc] + L.r_xi[3 * i + 1] * wu_s[3 * kc + 1] + L.r_xi[3 * i + 2] * wu_s[3 * kc + 2]) - L.r_b[i];
                    nrv += __popcll(__ballot(in && v > tol));
                }
                trivial = trivial && nrv == 0;
                const int key2 = (int)(key_soft + 1.5f * (float)nsat + 5.6f * (float)nrv);
#ifdef DMPC_DEV_TRACE   // development: the raw ingredients in the upper bits of the word (tools/gpu_key_features.py; order_kernel reads bits 0-8 only)
                const float smin = -(float)wave_max((double)-key_share);
                const int qs = smin >= 1.f ? 31 : (smin <= 0.f ? 0 : (int)(smin * 31.f));
                if (lane == 0) hdr[7] = (key2 > 255 ? 255 : key2) | (qs << 9) | ((nsat > 63 ? 63 : nsat) << 14) | ((nrv > 63 ? 63 : nrv) << 20) | ((ls > 7 ? 7 : ls) << 26);
#else
                if (lane == 0) hdr[7] = key2 > 255 ? 255 : key2;
#endif
            }
            if (trivial && !soft) {
                LSYNC();
                if (comp) { a_s[lane] = a_unc; wu_s[lane] = w_unc; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the rows this wave wrote are read back
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                LSYNC();
                bool rv = false;
                for (int i = lane; i < nr; i += 64) {
                    const int kc = L.r_kc[i];
                    const double v = -(L.r_xi[3 * i] * wu_s[3 * kc] + L.r_xi[3 * i + 1] * wu_s[3 * kc + 1] + L.r_xi[3 * i + 2] * wu_s[3 * kc + 2]) - L.r_b[i];
                    rv = rv || v > tol;
                }
                trivial = !__any(rv);
            }
            if (trivial) {
                int st = ST_SOLVED | (coll_flag ? ST_COLL : 0);
                const double p_out = w_unc + p0_l;
                double v_out = 0.0;
                if (comp) v_out = vel_out(a_s, k_l, ax_l, Qp->h, vo_l);
                const bool ob_check = !(var == VAR_ELLIP || var == VAR_SOFTALL || var == VAR_SOFTALL_C || var == VAR_SCP || var == VAR_CPP1 || cppv);   // (as the solver's output stage)
                if (ob_check) {
                    const double tolb = 50e-3;
                    bool bad = false;
                    if (lane < 3) bad = !(p_out < (lane == 0 ? Qp->pmax[0] : (lane == 1 ? Qp->pmax[1] : Qp->pmax[2])) + tolb) || !(p_out > (lane == 0 ? Qp->pmin[0] : (lane == 1 ? Qp->pmin[1] : Qp->pmin[2])) - tolb);
                    if (__any(bad)) st |= ST_OUTBOUND;
                }
                if (comp) {
                    Qp->p_out[(size_t)gid * N3 + lane] = p_out;
                    Qp->v_out[(size_t)gid * N3 + lane] = v_out;
                    Qp->a_out[(size_t)gid * N3 + lane] = a_unc;
                    if (Qp->lT_next) Qp->lT_next[(size_t)scene * N3 * C + cl + (size_t)(unsigned)(lane * C)] = p_out;
                }
                if (lane == 0) {
                    Qp->status[gid] = st;
                    hdr[4] = (violation ? 1 : 0) | (coll_flag ? 4 : 0) | 16;
                    hdr[7] = 256;   // (the order kernel reads this word only)
                    if (Qp->info) {
                        int *inf = Qp->info + (size_t)gid * 8;
                        inf[0] = viol_k; inf[1] = nrows_ref; inf[2] = 1; inf[3] = ccase;
                        inf[4] = 0; inf[5] = 0; inf[6] = 0; inf[7] = 0;
                    }
                }
            }
        }
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char dmpc_smem[];

#include "dmpc_solve.hip"
#include "dmpc_rsolve.hip"

// Scan phase: blockDim.x / 64 independent waves per workgroup, one agent each (P.lds_per_wave bytes of LDS per wave).
// Single-wave workgroups leave the launch bound by the workgroup dispatch rate (51 200 workgroups in ~180 us whatever the
// waves do); the waves never synchronise.
// XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, private L2 each), so they are
// renumbered such that every XCD works on whole scenes: a scene's prediction table is then fetched into ONE L2 instead of
// eight.  Pure performance remap (a bijection on [0, gridDim)).
template <bool SOFT, typename TT, bool FAST, bool ORD4 = false>
__global__ __launch_bounds__(64 * SCAN_WAVES_PER_WG, SCAN_WAVES_PER_SIMD) void dmpc_scan_kernel(StepParams P)
{
    const int W = (int)(blockDim.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7, y = (int)blockIdx.x >> 3;
    int off = 0;
    for (int xx = 0; xx < x; ++xx) off += (nb - xx + 7) >> 3;
    const int agent = (off + y) * W + wave, total = P.S * P.c_count;
    if (P.zero4 && blockIdx.x == 0 && threadIdx.x < 4) P.zero4[threadIdx.x] = 0;   // (nothing reads them before the order / solve kernels of this step)
    if (P.gzero) for (int z = (int)(blockIdx.x * blockDim.x + threadIdx.x); z < P.gzero_n; z += (int)(gridDim.x * blockDim.x)) P.gzero[z] = 0;   // (the cell grid's counters: their last reader, the list query, ran before this launch)
    if (agent >= total) return;
    scan_body<SOFT, TT, FAST, ORD4>(P, lane, agent, dmpc_smem + (size_t)wave * P.lds_per_wave);
}
// Solve phase, one agent per 64-thread workgroup (shallow launches: bound by their slowest agent)
template <bool SOFT, int QCAP, typename TF = double>
__global__ __launch_bounds__(64, SOLVE_WAVES_PER_SIMD) void dmpc_solve_kernel(StepParams P)
{
    int tk_unused = 0; bool cl_unused = false;
    solve_body<SOFT, QCAP, false, QCAP, TF>(P, threadIdx.x, blockIdx.x, gridDim.x, dmpc_smem, nullptr, false, tk_unused, cl_unused);
}

// solveDMPC.m:17-72, the SCP loop of ONE agent inside one launch: every pass is the scan (CheckCollDMPC + CollConstrDMPC about the previous
// pass's prediction, scan_body<.., SCP>) followed by the slack-free QP (solve_body) of the same wave; between the two and between passes the
// agent's rows, header and outputs travel through its slices of the global scratch / output arrays (same wave, same CU: workgroup-scope fences
// order them).  The loop ends when the pass's largest position change is <= tol (maxDeviation.m), after k_hor passes, or with the first
// infeasible pass (`success = 0`, :58-63).  Scan and solver use the same LDS one after the other.
__global__ __launch_bounds__(64, SOLVE_WAVES_PER_SIMD) void dmpc_scp_kernel(StepParams P)
{
    const int lane = (int)threadIdx.x;
    const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7, y = (int)blockIdx.x >> 3;
    int off = 0;
    for (int xx = 0; xx < x; ++xx) off += (nb - xx + 7) >> 3;
    const int vb = off + y;   // (solve_body renumbers blockIdx the same way)
    const int scene = vb / P.c_count, ci = vb - scene * P.c_count, gid = scene * P.c_count + ci;
    if (P.zero4 && blockIdx.x == 0 && threadIdx.x < 4) P.zero4[threadIdx.x] = 0;
    unsigned mask = 0u;
    double prev_l = 0.0;
    if (lane < N3) prev_l = P.lT[((size_t)(P.g_local * P.S + scene) * N3 + lane) * P.C + P.c_first + ci];   // prev_p = l(:,:,n) (:10)
    int passes = 0, iters_sum = 0, st = 0;
    const double *own_out = P.p_out + (size_t)gid * N3;
    for (int i = 1; i <= K; ++i) {   // `while (i <= k_hor && val > tol)` (:17); val = tol + 2 before the first pass
        scan_body<false, double, false, false, true>(P, lane, vb, dmpc_smem, i > 1 ? own_out : nullptr, &mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        LSYNC();
        if (UNI(P.hdr[(size_t)gid * 8 + 4]) & 8) return;   // scene of a transition that already stopped
        int tk_unused = 0; bool cl_unused = false;
        solve_body<false, 48, false>(P, lane, (int)blockIdx.x, nb, dmpc_smem, nullptr, false, tk_unused, cl_unused);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        LSYNC();
        ++passes;
        st = UNI(P.status[gid]);
        if (P.info) iters_sum += UNI(P.info[(size_t)gid * 8 + 4]);
        if (!(st & ST_SOLVED)) break;
        // val = maxDeviation(p, prev_p) (:69): maxDeviation.m:3 takes K = length(p)/3 of the 3 x k_hor matrix -- the first max(3, k_hor)/3 = 5 steps
        const double new_l = lane < N3 ? own_out[lane] : 0.0;
        const double dl = new_l - prev_l, d2 = dl * dl;
        // (lanes 3k: the squared distance of step k; the maximum over the first five steps through the fixed-order wave reduction)
        const double s3 = (d2 + dpp_row_shl<1>(d2)) + dpp_row_shl<2>(d2);
        const bool head = lane < 3 * ((K > 3 ? K : 3) / 3) && (lane % 3) == 0 && (lane & 15) <= 13;
        double val = wave_max(head ? sqrt(s3) : 0.0);
        {   // steps whose three lanes straddle a 16-lane row (step 5 = lanes 15..17 is beyond the five steps looked at: none here)
            static_assert(((K > 3 ? K : 3) / 3) * 3 <= 16, "maxDeviation: the steps looked at lie in the first DPP row");
        }
        prev_l = new_l;   // (:70)
        if (!(val > P.scp_tol)) break;
    }
    if (lane == 0 && P.info) { P.info[(size_t)gid * 8 + 2] = passes; P.info[(size_t)gid * 8 + 4] = iters_sum; }
}

// Persistent form of the REDUCED solver (dmpc_rsolve.hip): workgroups of RSOLVE_WAVES independent waves, as many per CU as the registers
// allow (no tables, 768 bytes of LDS per wave); the queue is the one of dmpc_solve_persist_kernel.
#ifndef RSOLVE_WAVES
#define RSOLVE_WAVES 8
#endif
constexpr int RSOLVE_LDS_PER_WAVE = 4 * 70 * 8;   // the output stage's 96 doubles; the ladder certificate's planes (4 doubles each: 64 rows + 6 box faces)
__global__ __launch_bounds__(RSOLVE_WAVES * 64) void dmpc_rsolve_persist_kernel(StepParams P)
{
    int total = P.S * P.c_count;
    if (P.live_bound) { const int lb = *P.live_bound; total = lb < total ? lb : total; }
    if (total == 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *mine = (unsigned char *)__builtin_assume_aligned(dmpc_smem + (size_t)wave * RSOLVE_LDS_PER_WAVE, 16);
    const int nw = (int)(gridDim.x * (blockDim.x >> 6));
    const int rest = total > nw ? total - nw : 0;
    const int T1 = rest < QUEUE_T1 * nw ? rest : QUEUE_T1 * nw;
    const int T3 = (rest - T1) < QUEUE_T3 * nw ? (rest - T1) : QUEUE_T3 * nw;
    const int CHUNK = P.queue_chunk > 0 ? P.queue_chunk : QUEUE_CHUNK;
    const int mid = rest - T1 - T3, T2 = (mid + CHUNK - 1) / CHUNK;
    const bool dyn = P.counter != nullptr;
    auto resolve = [&](int ps) -> int { return (ps < total && P.order) ? P.order[ps] : ps; };
    auto decode = [&](int t, int &left) -> int {
        left = 0;
        if (t < T1) return nw + t;
        if (t < T1 + T2) {
            const int ps = nw + T1 + CHUNK * (t - T1), end = nw + T1 + mid;
            left = (end - ps < CHUNK ? end - ps : CHUNK) - 1;
            return ps;
        }
        return nw + T1 + mid + (t - T1 - T2);
    };
    int pos = wave * (int)gridDim.x + (int)blockIdx.x, left = 0;
    for (;;) {
        if (pos >= total) break;
        int tkv = 0;
        bool claimed = false;
        const bool want = dyn && left == 0;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int agent = resolve(pos);
#ifdef RSOLVE_TRACE
        const long long t_a = wall_clock64();
#endif
        rsolve_body(P, ln, agent, mine, want, tkv, claimed);
        if (want && !claimed && lane == 0) tkv = atomicAdd(P.counter, 1);
        LSYNC();
#ifdef RSOLVE_TRACE
        // development: start time and duration of every agent (dmpc_debug_trace with agent = -5), 100 MHz ticks
        if (P.dbg && P.dbg_agent == -5 && lane == 0) { P.dbg[(size_t)agent * 2] = (double)t_a; P.dbg[(size_t)agent * 2 + 1] = (double)(wall_clock64() - t_a); }
#endif
        if (!dyn) { pos += nw; continue; }
        if (left > 0) { pos++; left--; continue; }
        pos = decode(__builtin_amdgcn_readfirstlane(tkv), left);
    }
}

// Persistent form of the solve phase: one workgroup of up to 8 independent waves per CU (two per SIMD).  The Gram tables of the three
// cost cases and the Lambda table (23 KB) are staged in LDS ONCE per workgroup and shared by its waves; every wave then
// claims agents from a global queue (heaviest first when order_kernel ran) until it is empty, so a wave that finishes a
// light agent immediately starts the next one and the launch ends when the LAST agent ends, not when the slowest
// workgroup slot drains.  No workgroup barrier after the table load: the waves never synchronise.
// (slack-free variants: 9 waves per workgroup -- with the row flags as bits a ninth wave fits next to the tables in the CU's 160 KB,
// and 168 registers per lane hold the kernel without spills; the slack variants need 17 KB of LDS and 230 registers per wave: 8)
// (round 4, slack-free variants: TWELVE waves per workgroup -- three per SIMD, what 168 registers per lane allow -- with the split T of
// dmpc_solve.hip: TS columns of the factor per wave and a pool of P.n_ext extensions behind the waves' blocks)
template <bool SOFT, int QCAP, int TS = QCAP, typename TF = double>
__global__ __launch_bounds__(SOFT ? 512 : DMPC_HARD_PW * 64, 1) void dmpc_solve_persist_kernel(StepParams P)
{
    // tier 2 works through the list of agents tier 1 flagged (P.order points at it); usually it is empty
    int total = P.only_flagged ? *P.flag_count : P.S * P.c_count;
    if (!P.only_flagged && P.live_bound) { const int lb = *P.live_bound; total = lb < total ? lb : total; }   // the rest of the order: agents the scan finished
    if (total == 0) return;
    double *shtab = (double *)dmpc_smem;
    for (int i = threadIdx.x; i < TAB_DOUBLES; i += blockDim.x) shtab[i] = P.tables[i];
    if (TS < QCAP) {   // the pool of T extensions: all free, all zero
        double *ex = (double *)(dmpc_smem + PERSIST_TABLE_BYTES + (size_t)(blockDim.x >> 6) * P.lds_per_wave);
        for (int i = threadIdx.x; i < P.n_ext * ext_doubles(QCAP, TS) + (int)(EXT_PAD_BYTES / 8); i += blockDim.x) ex[i] = 0.0;
        if (threadIdx.x == 0) *(unsigned *)(dmpc_smem + PERSIST_TABLE_BYTES - 16) = (1u << P.n_ext) - 1u;
    }
    __syncthreads();
    // readfirstlane: tells the compiler the wave index (and with it every LDS base address and every value read
    // through one) is wave-uniform -- otherwise the solver's uniform branches are compiled as divergent ones
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *mine = (unsigned char *)__builtin_assume_aligned(dmpc_smem + PERSIST_TABLE_BYTES + (size_t)wave * P.lds_per_wave, 16);
    // The queue.  First round: position wave * #workgroups + workgroup, i.e. the heaviest #waves agents each get a wave (and
    // the heaviest #workgroups a CU) of their own.  Later positions are claimed with TICKETS from one global counter.  An
    // atomic on one address costs ~12 ns of serialised service on this part (51 200 single claims = 0.6 ms, measured: the
    // launch could not end sooner whatever the solver did), and its result takes microseconds to arrive; so
    //   * a ticket stands for one position where the agents are heavy (the #waves positions after the first round) and
    //     where the launch ends (the last #waves positions: a fine-grained tail), and for QUEUE_CHUNK adjacent positions in
    //     between (the light bulk): fewer atomics, at most one light chunk of imbalance -- and a pre-claimed chunk waits behind
    //     its wave's current agent, which is why the chunk is short;
    //   * the next ticket is claimed when the current agent is as good as solved (solve_body: CLAIM_NEXT) and read after its output
    //     stage: most of the latency hides there, and no position waits behind a solve that turns out long.
    const int nw = (int)(gridDim.x * (blockDim.x >> 6));
    const int rest = total > nw ? total - nw : 0;
    const int T1 = rest < QUEUE_T1 * nw ? rest : QUEUE_T1 * nw;
    const int T3 = (rest - T1) < QUEUE_T3 * nw ? (rest - T1) : QUEUE_T3 * nw;
    // (round 5: per launch -- two positions per ticket pay where the queue is deep and its agents light, the 512-scene replays: fewer atomics;
    // in a shallow queue of heavy agents, the 10^4-agent scene at 5.6 agents per wave, the second position waits behind a solve that may turn
    // out long and such positions ended the launch: 0.75 -> 0.71 ms with single positions)
    const int CHUNK = P.queue_chunk > 0 ? P.queue_chunk : QUEUE_CHUNK;
    const int mid = rest - T1 - T3, T2 = (mid + CHUNK - 1) / CHUNK;
    // (Rounds 2-3 claimed the next ticket right BEFORE the current agent was solved -- the atomic's latency hides completely -- and the
    // positions parked behind long solves ended the launch late: waves ending 779-878 us.  Claiming two positions ahead -- the agent behind the next ticket resolved through the order during the solve, so that
    // only one memory round trip per agent is exposed -- was built and measured in round 3: 0.935 against 0.891 ms on the headline
    // launch.  A claimed position waits behind its wave's current agent, and when that one is a 300-500 us infeasibility proof the
    // light agents parked behind it end the launch late.)
    const bool dyn = P.counter != nullptr;
    auto resolve = [&](int ps) -> int { return (ps < total && P.order) ? P.order[ps] : ps; };   // queue position -> agent
    auto decode = [&](int t, int &left) -> int {   // ticket -> first position it stands for (+ `left` further ones)
        left = 0;
        if (t < T1) return nw + t;
        if (t < T1 + T2) {
            const int ps = nw + T1 + CHUNK * (t - T1), end = nw + T1 + mid;
            left = (end - ps < CHUNK ? end - ps : CHUNK) - 1;
            return ps;
        }
        return nw + T1 + mid + (t - T1 - T2);
    };
    int pos = wave * (int)gridDim.x + (int)blockIdx.x, left = 0;
#ifdef DMPC_DEV_TRACE
    // development: start / end time and agent count of every wave (dmpc_debug_trace with agent = -2)
    const long long t_begin = wall_clock64();
    int n_done = 0;
#endif
    for (;;) {
        if (pos >= total) break;
#ifdef DMPC_DEV_TRACE
        n_done++;
        const long long t_a = wall_clock64();
#endif
        int tkv = 0;
        bool claimed = false;
        const bool want = dyn && left == 0;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int agent = resolve(pos);
        solve_body<SOFT, QCAP, true, TS, TF>(P, ln, agent, total, mine, shtab, want, tkv, claimed,
                                         (int)(((blockDim.x >> 6) - wave) * (P.lds_per_wave >> 3)));
        if (want && !claimed && lane == 0) tkv = atomicAdd(P.counter, 1);   // (the quick ways out of the solver: stopped scene, agent finished by the scan)
        LSYNC();
#ifdef DMPC_DEV_TRACE
        // development: start time and duration of every queue position (dmpc_debug_trace with agent = -3) / of every agent (-5)
        if (P.dbg && P.dbg_agent == -3 && lane == 0) { P.dbg[(size_t)pos * 2] = (double)t_a; P.dbg[(size_t)pos * 2 + 1] = (double)(wall_clock64() - t_a); }
        if (P.dbg && P.dbg_agent == -5 && lane == 0) { P.dbg[(size_t)agent * 2] = (double)t_a; P.dbg[(size_t)agent * 2 + 1] = (double)(wall_clock64() - t_a); }
#endif
        if (!dyn) { pos += nw; continue; }
        if (left > 0) { pos++; left--; continue; }
        pos = decode(__builtin_amdgcn_readfirstlane(tkv), left);
    }
#ifdef DMPC_DEV_TRACE
    if (P.dbg && P.dbg_agent == -2 && lane == 0) {
        double *d = P.dbg + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 3;
        d[0] = (double)t_begin; d[1] = (double)wall_clock64(); d[2] = (double)n_done;
    }
#endif
}

// --------------------------------------------------------------------------------------------
// small layout / bookkeeping kernels
// --------------------------------------------------------------------------------------------

// Axis-aligned bounding boxes of every agent's predicted horizon, one per third of the horizon (steps 0-4, 5-9, 10-14):
// bbox[G][S][18][C] (fp32, rounded outwards) = per segment xmin,xmax,ymin,ymax,zmin,zmax.  A neighbour can come within R of the agent at step k only
// if the boxes of the segment that holds k are within R per axis.  (One box for the whole horizon lets 20 % of a 10^4-agent
// scene through once the agents move -- 4.3 m of horizon each; the three segments 4-5 times fewer.)
constexpr int NSEG = 3, SEG_STEPS = K / NSEG;
static_assert(NSEG * SEG_STEPS == K, "horizon segments");
template <typename TT>
__global__ void bbox_kernel(int total, int C, const TT *__restrict__ lT, float *__restrict__ bbox, float *__restrict__ bbox_nm /* [total][NBOX_NM]: the same boxes, one neighbour's 18 numbers contiguous (scalar loads of nbr_kernel) */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // enumerates (g*S + s)*C + c
    if (i >= total) return;
    const int c = i % C;
    const size_t gs = (size_t)(i / C);
    const TT *src = lT + gs * N3 * C + c;
    float *dst = bbox + gs * (6 * NSEG) * C + c;   // fp32, rounded outwards: the boxes only ever have to be conservative
    for (int sg = 0; sg < NSEG; ++sg) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int k = sg * SEG_STEPS; k < (sg + 1) * SEG_STEPS; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double v = (double)src[(size_t)(3 * k + a) * C];
                lo[a] = fmin(lo[a], v); hi[a] = fmax(hi[a], v);
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float flo = __double2float_rd(lo[a]), fhi = __double2float_ru(hi[a]);
            dst[(size_t)(6 * sg + 2 * a) * C] = flo; dst[(size_t)(6 * sg + 2 * a + 1) * C] = fhi;
            bbox_nm[(size_t)i * NBOX_NM + 6 * sg + 2 * a] = flo; bbox_nm[(size_t)i * NBOX_NM + 6 * sg + 2 * a + 1] = fhi;
        }
    }
}

// Neighbour-major fp32 copy of the prediction table for the list walk of the scan: out[chunk][scene][column][64] with element
// 4k + axis = component (k, axis) of that neighbour's horizon, 0 in the fourth slot of every step and in the last four.
template <typename TT>
__global__ void table_nbrmajor_kernel(size_t total, int C, const TT *__restrict__ lT, float *__restrict__ out, int *__restrict__ zbuf /* or null: the cell-grid counters, zeroed here */, size_t nz)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t z = t; z < nz; z += (size_t)gridDim.x * blockDim.x) zbuf[z] = 0;
    if (t >= total) return;
    const int l = (int)(t & 63), k4 = l >> 2, a4 = l & 3;
    const size_t gc = t >> 6, gs = gc / (size_t)C, c = gc - gs * (size_t)C;
    out[t] = (a4 < 3 && k4 < K) ? (float)lT[(gs * N3 + 3 * k4 + a4) * C + c] : 0.f;
}

// Neighbour lists of large scenes: the all-pairs test of the segment boxes.  LANES = 64 AGENTS of a scene (their inflated boxes in
// registers), the neighbours are walked one by one with the neighbour's 18 box numbers in SGPRs (five 16-byte scalar loads from the
// neighbour-major copy of the boxes): 18 compares per neighbour test 64 agent-neighbour pairs, nothing goes through LDS.  A neighbour
// whose box overlaps an agent's in any segment is appended to that agent's list -- per-lane counters, so every list is in increasing
// neighbour order.  (Round 2 had the roles the other way round -- a tile of 64 neighbours in the lanes, 8 agents' boxes broadcast from
// LDS, 144 LDS reads per tile and wave: at N = 10^4 that is 2.8e7 LDS instructions through ONE LDS pipe per CU, 0.2-0.37 ms whatever
// the VALU did.)
// code = (chunk << 20) | column, as the scan decodes it.  The neighbours of a scene are split over NBR_PARTS waves per agent block
// (N / 64 waves alone do not fill the chip); part q writes its survivors to the q-th piece of every agent's list (cap / NBR_PARTS
// entries) and their number to cnt_out[agent][q] (-1: did not fit: the scan walks the whole table); the scan kernel closes the gaps.
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef const f4_t __attribute__((address_space(4))) *ConstF4;
__global__ __launch_bounds__(64) void nbr_kernel(int S, int G, int C, int g_local, int c_first, int c_count, int short_from, float R, float Rz,
                                                 const float *__restrict__ bbox, const float *__restrict__ bbox_nm, int cap, int *__restrict__ list,
                                                 int *__restrict__ cnt_out)
{
    constexpr int NB = 6 * NSEG;
    const int nblk = (c_count + 63) >> 6;
    const int part = blockIdx.x % NBR_PARTS, sb = blockIdx.x / NBR_PARTS;
    const int scene = sb / nblk, b = sb - scene * nblk;
    const int lane = threadIdx.x;
    const int ci = b * 64 + lane;
    const bool mine = ci < c_count;
    const int cl = c_first + (mine ? ci : 0);
    // own boxes, inflated: even entries lower bound - R, odd entries upper bound + R (R carries a 1e-4 margin: far above the fp32
    // rounding of these sums); z by Rz (the metric's z scale)
    float ob[NB];
#pragma unroll
    for (int x = 0; x < NB; ++x) {
        const float v = bbox[((size_t)(g_local * S + scene) * NB + x) * C + cl];
        const float infl = (x % 6 < 4) ? R : Rz;
        ob[x] = (x & 1) ? v + infl : v - infl;
    }
    const int pcap = cap / NBR_PARTS;
    const size_t gid = (size_t)scene * c_count + (mine ? ci : 0);
    int *mylist = list + gid * (size_t)cap + (size_t)part * pcap;
    const long E = (long)G * C;
    const int e_lo = (int)(E * part / NBR_PARTS), e_hi = (int)(E * (part + 1) / NBR_PARTS);
    int cnt = 0;
    // the overlap test in arithmetic form: two boxes overlap on an axis iff max(n_lo - o_hi, o_lo - n_hi) <= 0; a segment overlaps iff the
    // maximum over its six differences is <= 0, a neighbour is listed iff the minimum over the three segments is.  6 subtractions + 3
    // three-operand maxima per segment with the neighbour's numbers as scalar operands: no mask logic on the scalar unit.  (A difference
    // that rounds or flushes to zero reads as "touching": the list only ever grows by that, it is a superset by design.)
    for (int e = e_lo; e < e_hi;) {
        const int r = e / C, j_lo = e - r * C;
        const int j_hi = (e_hi - e) < (C - j_lo) ? j_lo + (e_hi - e) : C;                      // this part's columns of chunk r
        const int j_end = j_hi < C - ((short_from && r >= short_from) ? 1 : 0) ? j_hi : C - ((short_from && r >= short_from) ? 1 : 0);   // (padding column of a short chunk)
        const float *base = bbox_nm + ((size_t)(r * S + scene) * C) * NBOX_NM;
        const int self = (r == g_local) ? cl : -1;
        for (int jc = j_lo; jc < j_end; ++jc) {
            const ConstF4 nb = (ConstF4)(unsigned long long)(base + (size_t)jc * NBOX_NM);
            const f4_t n0 = nb[0], n1 = nb[1], n2 = nb[2], n3 = nb[3], n4 = nb[4];
            const float nv[20] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w, n4.x, n4.y, n4.z, n4.w};
            float sgm[NSEG];
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                const float a0 = nv[6 * sg] - ob[6 * sg + 1], a1 = ob[6 * sg] - nv[6 * sg + 1], a2 = nv[6 * sg + 2] - ob[6 * sg + 3];
                const float a3 = ob[6 * sg + 2] - nv[6 * sg + 3], a4 = nv[6 * sg + 4] - ob[6 * sg + 5], a5 = ob[6 * sg + 4] - nv[6 * sg + 5];
                sgm[sg] = fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), a2), __builtin_fmaxf(__builtin_fmaxf(a3, a4), a5));
            }
            const float worst = __builtin_fminf(__builtin_fminf(sgm[0], sgm[1]), sgm[2]);
            if (worst <= 0.f && mine && jc != self) {
                if (cnt < pcap) mylist[cnt] = (r << 20) | jc;
                cnt++;
            }
        }
        e += j_hi - j_lo;
    }
    if (mine) cnt_out[gid * NBR_PARTS + part] = cnt > pcap ? -1 : cnt;
}

// --------------------------------------------------------------------------------------------
// Neighbour lists of large scenes, round 4: a cell grid instead of the all-pairs box test, and the fp32 DISTANCE test of the scan's list
// walk moved in here.  (C4, 10^4 agents, round 3: nbr_kernel is an O(N^2) box test -- 222 us -- whose lists hold the 100-700 neighbours
// whose horizon boxes come close; every lane appended to its own list at a 16 KB stride, 137 MB of written cache lines for 12 MB of
// entries; the scan then walked those lists once per agent with a whole horizon of distances per entry, 327 us.)
//   grid_bin / grid_scan / grid_fill: the agents of a scene binned by the CENTRE of their whole-horizon box (counting sort: count,
//     exclusive scan per scene, scatter; the order inside a cell is whatever the atomics give -- nothing below depends on it);
//   grid_query: one workgroup per agent, one wave per horizon segment.  Candidates = the entries of the cells within own half extent + R +
//     the scene's LARGEST half extent of the own box centre (a running maximum from grid_bin), streamed as two coalesced 16-byte halves
//     per candidate, lanes = candidates; pre-test = closest approach of the two segment CHORDS at equal time against R + both deviations
//     from the chord (grid_fill_kernel; the segment-box test of nbr_kernel passed 609 candidates per agent at N = 10^4, this one 157);
//     survivors compacted into an LDS staging list; full waves of survivors then take the distance test itself -- the candidate's segment
//     from the fp32 neighbour-major table (5 x 16 bytes per lane) against the own one (LDS broadcasts), pass = some horizon step closer
//     than the selection radius (3 rmin; 1 for the hard rows), widened by 1e-3 as in the scan -- and set their bit in the workgroup's
//     BITMAP over the scene's agents; the list is the bitmap read in order: increasing neighbour index whatever order the candidates came
//     in (the scan builds rows in list order = the reference's row order), duplicates impossible, written as one contiguous run.  Lists of
//     10-40 entries (100 at N = 10^4) instead of 100-700: a superset of every pair the scan can select at any step, so results are
//     unchanged bit for bit (tests/test_gpu_paths.py against option no_cull).
//     N = 10^4, one scene: 171 us with the box pre-test and 32-byte records -> 106 (chord pre-test) -> 94 (records as two arrays: a wave's
//     load is whole cache lines) -> 91 (agents taken in cell order) -> 88 us (a wave per segment).  What is left is the candidate stream,
//     2 046 records per agent = 655 MB per step out of the L2s, at ~80 instructions per round of 64: without any survivor the kernel took
//     90 of 106 us, without the stream 18 (both compiled out); its loads were NOT in flight during the previous round's test until the
//     instruction stream was read (see the loop), and putting them in flight changed nothing -- throughput, not latency.
// --------------------------------------------------------------------------------------------
struct GridGeom {
    float org[3], inv[3];   // cell index along axis a = clamp(floor((x - org[a]) * inv[a]), 0, n[a] - 1)
    int n[3];
};
__device__ __forceinline__ int grid_coord(const GridGeom &g, int a, float x)
{
    const float t = (x - g.org[a]) * g.inv[a];
    const int c = (int)floorf(t);
    return c < 0 ? 0 : (c >= g.n[a] ? g.n[a] - 1 : c);   // clamping is monotone: points within d of each other stay within ceil(d / cell) cells
}
// thread per table column (chunk, scene, column): per horizon segment the cell of the segment box's centre, cell counts, the scene's
// largest half extents (one atomic per wave and quantity when the wave's columns belong to one scene -- 10^4 same-address atomics are
// 0.1 ms of serialised service each)
__global__ void grid_bin_kernel(int total, int S, int C, int short_from, GridGeom gg, const float *__restrict__ bbox_nm, int *__restrict__ cellof,
                                int *__restrict__ cnt, int *__restrict__ maxhalf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (r * S + scene) * C + c
    const bool in = i < total;
    const int ii = in ? i : total - 1;
    const int c = ii % C, rs = ii / C, scene = rs % S, r = rs / S;
    const bool valid = in && !(short_from && r >= short_from && c == C - 1);   // (padding column of a short chunk)
    const float *b = bbox_nm + (size_t)ii * NBOX_NM;
    const int ncell = gg.n[0] * gg.n[1] * gg.n[2];
    const int sc0 = __builtin_amdgcn_readfirstlane(scene);
    const bool uni = __all(scene == sc0);
    // (blockIdx.y = segment: a thread's work is a chain of dependent round trips -- loads, wave maxima, atomics -- and 10^4 threads do not fill
    // the chip: three times as many, a third as long)
    {
        const int sg = (int)blockIdx.y;
        int cc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = b[6 * sg + 2 * a], hi = b[6 * sg + 2 * a + 1];
            const float half = 0.5f * (hi - lo) * 1.0001f + 1e-5f;   // (rounded up: the query's reach must cover it)
            cc[a] = grid_coord(gg, a, 0.5f * (lo + hi));
            int hb = valid ? __float_as_int(half) : 0;   // non-negative floats order like integers
            int *dst = maxhalf + ((size_t)scene * NSEG + sg) * 3 + a;
            if (uni) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(hb, off); hb = o > hb ? o : hb; }
                if ((threadIdx.x & 63) == 0 && hb > *(volatile int *)dst) atomicMax(dst, hb);
            } else if (valid && hb > *(volatile int *)dst) atomicMax(dst, hb);
        }
        const int cell = (cc[2] * gg.n[1] + cc[1]) * gg.n[0] + cc[0];
        if (in) cellof[(size_t)sg * total + i] = valid ? cell : -1;
        if (valid) atomicAdd(cnt + ((size_t)scene * NSEG + sg) * ncell + cell, 1);
    }
}
// block per (scene, segment): start[..][0 .. ncell] = exclusive prefix of the cell counts; the counts are zeroed (grid_fill counts them up again)
__global__ void grid_scan_kernel(int ncell, int *__restrict__ cnt, int *__restrict__ start)
{
    __shared__ int part[1024];   // (blockDim.x <= 1024 threads per chunk: 1 560 cells at N = 10^4 are two chunks)
    __shared__ int carry;
    const int scene = blockIdx.x, t = threadIdx.x, nt = (int)blockDim.x;
    int *c = cnt + (size_t)scene * ncell, *st = start + (size_t)scene * (ncell + 1);
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += nt) {
        const int v = base + t < ncell ? c[base + t] : 0;
        part[t] = v;
        __syncthreads();
        for (int off = 1; off < nt; off <<= 1) {
            const int add = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        if (base + t < ncell) { st[base + t] = carry + part[t] - v; c[base + t] = 0; }
        __syncthreads();
        if (t == 0) carry += part[nt - 1];
        __syncthreads();
    }
    if (t == 0) st[ncell] = carry;
}
// entry = 32-byte record: the query streams its candidates COALESCED (a gather of the candidates' data -- 64 cache lines per wave load --
// kept the address units busy for the whole kernel).  What a record holds is the segment's CHORD: {code, first step x, y, z/c, last - first step
// x, y, z/c, largest distance of the segment's steps from the chord at their own time} -- the query's pre-test is the closest approach of
// two chords AT EQUAL TIME (two agents whose segment boxes overlap are usually at the overlap at different times: the box test let 609
// candidates per agent through to the distance test at N = 10^4, the chord test 157; 100 are listed)
__global__ void grid_fill_kernel(int total, int S, int C, int ncell, float e1z, const int *__restrict__ cellof, int *__restrict__ cnt, const int *__restrict__ start,
                                 const float *__restrict__ lrow, f4_t *__restrict__ ent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = i % C, rs = i / C, scene = rs % S, r = rs / S;
    const size_t nag = (size_t)total / S;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int cell = cellof[(size_t)sg * total + i];
        if (cell < 0) return;
        const size_t ss = (size_t)scene * NSEG + sg;
        const int pos = atomicAdd(cnt + ss * ncell + cell, 1);
        const f4_t *row = (const f4_t *)lrow + (size_t)i * 16 + SEG_STEPS * sg;
        f4_t v[SEG_STEPS];
#pragma unroll
        for (int u = 0; u < SEG_STEPS; ++u) { v[u] = row[u]; v[u].z *= e1z; }
        const f4_t a0 = v[0], a1 = v[SEG_STEPS - 1];
        float dev2 = 0.f;
#pragma unroll
        for (int u = 1; u < SEG_STEPS - 1; ++u) {
            const float t = (float)u / (float)(SEG_STEPS - 1);
            const float dx = v[u].x - (a0.x + t * (a1.x - a0.x)), dy = v[u].y - (a0.y + t * (a1.y - a0.y)), dz = v[u].z - (a0.z + t * (a1.z - a0.z));
            dev2 = fmaxf(dev2, dx * dx + dy * dy + dz * dz);
        }
        const size_t ei = ss * nag + start[ss * (ncell + 1) + cell] + pos;
        f4_t r0, r1;
        r0.x = __int_as_float((r << 20) | c); r0.y = a0.x; r0.z = a0.y; r0.w = a0.z;
        r1.x = a1.x - a0.x; r1.y = a1.y - a0.y; r1.z = a1.z - a0.z; r1.w = sqrtf(dev2) * 1.0001f + 1e-5f;   // (rounded up)
        ent[ei] = r0; ent[(size_t)S * NSEG * nag + ei] = r1;   // two arrays of 16-byte halves: a wave's load of either is whole cache lines
    }
}
// ---- the cell grid of ONE scene in two launches (round 6; the five kernels above -- boxes, neighbour-major copy, bin, scan, fill: 42 us of
// a 0.7 ms MPC step at N = 10^4, every one of them a chain of dependent round trips at the launch floor -- stay for batches of scenes).
// grid_prep_kernel: blocks [0, nbA): thread per table column -- segment boxes (bbox_kernel), cell of each segment's centre, its position within the
// cell (the value the count's atomicAdd returns: the fill needs no second round of atomics), the scene's largest half extents through a
// block-wide maximum in LDS (nine global atomics per block); blocks [nbA, ..): the neighbour-major fp32 copy (table_nbrmajor_kernel).  The
// counters must be ZERO at entry: the scan kernel of the previous step leaves them so (StepParams::gzero), a memset the first time.
template <typename TT>
__global__ __launch_bounds__(256) void grid_prep_kernel(int total, int C, int short_from, GridGeom gg, int nbA, const TT *__restrict__ lT, float *__restrict__ bbox,
                                                        float *__restrict__ bbox_nm, float *__restrict__ lrow, int *__restrict__ cellof, int *__restrict__ posof,
                                                        int *__restrict__ cnt, int *__restrict__ maxhalf)
{
    if ((int)blockIdx.x >= nbA) {
        const size_t t = (size_t)((int)blockIdx.x - nbA) * 256 + threadIdx.x;
        if (t >= (size_t)total * 64) return;
        const int l = (int)(t & 63), k4 = l >> 2, a4 = l & 3;
        const size_t gc = t >> 6, gs = gc / (size_t)C, c = gc - gs * (size_t)C;
        lrow[t] = (a4 < 3 && k4 < K) ? (float)lT[(gs * N3 + 3 * k4 + a4) * C + c] : 0.f;
        return;
    }
    __shared__ int mh[NSEG * 3];
    if (threadIdx.x < NSEG * 3) mh[threadIdx.x] = 0;
    __syncthreads();
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;   // r * C + c (one scene)
    if (i < total) {
        const int c = i % C, r = i / C;
        const bool valid = !(short_from && r >= short_from && c == C - 1);   // (padding column of a short chunk)
        const TT *src = lT + (size_t)r * N3 * C + c;
        float *dst = bbox + (size_t)r * (6 * NSEG) * C + c;
        const int ncell = gg.n[0] * gg.n[1] * gg.n[2];
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
#pragma unroll
            for (int k = sg * SEG_STEPS; k < (sg + 1) * SEG_STEPS; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double v = (double)src[(size_t)(3 * k + a) * C];
                    lo[a] = fmin(lo[a], v); hi[a] = fmax(hi[a], v);
                }
            int cc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float flo = __double2float_rd(lo[a]), fhi = __double2float_ru(hi[a]);
                dst[(size_t)(6 * sg + 2 * a) * C] = flo; dst[(size_t)(6 * sg + 2 * a + 1) * C] = fhi;
                bbox_nm[(size_t)i * NBOX_NM + 6 * sg + 2 * a] = flo; bbox_nm[(size_t)i * NBOX_NM + 6 * sg + 2 * a + 1] = fhi;
                const float half = 0.5f * (fhi - flo) * 1.0001f + 1e-5f;   // (as grid_bin_kernel: rounded up, the query's reach must cover it)
                cc[a] = grid_coord(gg, a, 0.5f * (flo + fhi));
                if (valid) atomicMax(&mh[3 * sg + a], __float_as_int(half));   // non-negative floats order like integers
            }
            const int cell = (cc[2] * gg.n[1] + cc[1]) * gg.n[0] + cc[0];
            cellof[(size_t)sg * total + i] = valid ? cell : -1;
            posof[(size_t)sg * total + i] = valid ? atomicAdd(cnt + (size_t)sg * ncell + cell, 1) : 0;
        }
    }
    __syncthreads();
    if (threadIdx.x < NSEG * 3) { const int v = mh[threadIdx.x]; if (v > *(volatile int *)(maxhalf + threadIdx.x)) atomicMax(maxhalf + threadIdx.x, v); }
}
// grid_fill2_kernel: every block forms the exclusive prefix of the three segments' cell counts in its LDS (1 560 cells each at N = 10^4: cheaper
// than a launch of its own and a trip through memory), block 0 also writes it out for the query; then the entry records as grid_fill_kernel,
// at start[cell] + the position grid_prep_kernel drew.  Dynamic LDS: NSEG * (ncell + 1) ints.
__global__ __launch_bounds__(256) void grid_fill2_kernel(int total, int C, int ncell, float e1z, const int *__restrict__ cellof, const int *__restrict__ posof,
                                                         const int *__restrict__ cnt, int *__restrict__ start, const float *__restrict__ lrow, f4_t *__restrict__ ent)
{
    int *st = (int *)dmpc_smem;   // [NSEG][ncell + 1]
    __shared__ int wtot[NSEG][4];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = (ncell + 255) / 256;
    const int lo = t * chunk, hi = lo + chunk < ncell ? lo + chunk : ncell;
    // the three segments' prefixes together: a thread sums its chunk of each, one scan inside the wave (shuffles), the four waves' totals through LDS
    int sum[NSEG], inc[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int *c = cnt + (size_t)sg * ncell;
        int v = 0;
        for (int j = lo; j < hi; ++j) v += c[j];
        sum[sg] = v; inc[sg] = v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) { const int o = __shfl_up(inc[sg], off); if (lane >= off) inc[sg] += o; }
    if (lane == 63)
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) wtot[sg][wave] = inc[sg];
    __syncthreads();
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int *c = cnt + (size_t)sg * ncell;
        int run = inc[sg] - sum[sg];   // exclusive inside the wave
        for (int w = 0; w < wave; ++w) run += wtot[sg][w];
        for (int j = lo; j < hi; ++j) { st[sg * (ncell + 1) + j] = run; run += c[j]; }
        if (t == 255) st[sg * (ncell + 1) + ncell] = wtot[sg][0] + wtot[sg][1] + wtot[sg][2] + wtot[sg][3];
    }
    __syncthreads();
    if (blockIdx.x == 0) for (int j = t; j < NSEG * (ncell + 1); j += 256) start[j] = st[j];
    const int i = (int)blockIdx.x * 256 + t;
    if (i >= total) return;
    const int code = ((i / C) << 20) | (i % C);   // (chunk << 20) | column
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int cell = cellof[(size_t)sg * total + i];
        if (cell < 0) return;
        const int pos = posof[(size_t)sg * total + i];
        const f4_t *row = (const f4_t *)lrow + (size_t)i * 16 + SEG_STEPS * sg;
        f4_t v[SEG_STEPS];
#pragma unroll
        for (int u = 0; u < SEG_STEPS; ++u) { v[u] = row[u]; v[u].z *= e1z; }
        const f4_t a0 = v[0], a1 = v[SEG_STEPS - 1];
        float dev2 = 0.f;
#pragma unroll
        for (int u = 1; u < SEG_STEPS - 1; ++u) {
            const float tt = (float)u / (float)(SEG_STEPS - 1);
            const float dx = v[u].x - (a0.x + tt * (a1.x - a0.x)), dy = v[u].y - (a0.y + tt * (a1.y - a0.y)), dz = v[u].z - (a0.z + tt * (a1.z - a0.z));
            dev2 = fmaxf(dev2, dx * dx + dy * dy + dz * dz);
        }
        const size_t ei = (size_t)sg * total + st[sg * (ncell + 1) + cell] + pos;
        f4_t r0, r1;
        r0.x = __int_as_float(code); r0.y = a0.x; r0.z = a0.y; r0.w = a0.z;
        r1.x = a1.x - a0.x; r1.y = a1.y - a0.y; r1.z = a1.z - a0.z; r1.w = sqrtf(dev2) * 1.0001f + 1e-5f;   // (rounded up)
        ent[ei] = r0; ent[(size_t)NSEG * total + ei] = r1;
    }
}
// A workgroup = ONE agent, a wave per horizon segment (the segments are independent up to the bitmap, which the three waves share): three
// times as many, shorter waves -- the launch is a little more than one round of resident waves deep (39 agents per CU at N = 10^4 against
// 28 wave slots), and ends with the last agents' whole chains otherwise.
constexpr int GQ_WAVES = NSEG;
constexpr int GQ_STAGE = 128;      // staging list of box-test survivors per wave (a full wave is taken off it as soon as there is one)
inline size_t grid_query_lds(int nagents) { return (((size_t)((nagents + 31) / 32) * 4 + GQ_WAVES * GQ_STAGE * 4 + 15) & ~(size_t)15) + 16 * 16; }
typedef float f2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64 * GQ_WAVES) void grid_query_kernel(int S, int G, int C, int g_local, int c_first, int c_count, GridGeom gg, float R, float Rz, float e1z, float thr2,
                                                                   const float *__restrict__ bbox_nm, const float *__restrict__ lrow, const int *__restrict__ start,
                                                                   const f4_t *__restrict__ ent, const int *__restrict__ maxhalf, int cap, int cell_order,
                                                                   int *__restrict__ list, int *__restrict__ cnt_out)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int gid = (int)blockIdx.x;
    const int nag = G * C, nwords = (nag + 31) >> 5, ncell = gg.n[0] * gg.n[1] * gg.n[2];
    const int scene = gid / c_count;
    int ci = gid - scene * c_count;
    // cell_order (a query over whole scenes): wave number -> agent through the first segment's entry array, which IS the scene's agents sorted
    // by cell -- the waves of a workgroup, and of a CU, then stream the same cells' candidates (one trip to the L2 for the lot; in agent
    // order every wave fetched its 65 KB of candidates on its own: 655 MB per step at N = 10^4, 11 TB/s out of the L2s)
    if (cell_order) ci = __builtin_amdgcn_readfirstlane(__float_as_int(ent[(size_t)scene * NSEG * nag + ci].x) & 0xfffff);
    const int cl = c_first + ci;
    const int oid = scene * c_count + ci;   // where this agent's list goes
    unsigned *bits = (unsigned *)dmpc_smem;
    int *stage = (int *)(dmpc_smem + (size_t)nwords * 4) + wave * GQ_STAGE;
    f4_t *ownrow = (f4_t *)(dmpc_smem + (((size_t)nwords * 4 + GQ_WAVES * GQ_STAGE * 4 + 15) & ~(size_t)15));
    for (int i = (int)threadIdx.x; i < nwords; i += 64 * GQ_WAVES) bits[i] = 0u;
    const size_t self_i = (size_t)(g_local * S + scene) * C + cl;
    if (threadIdx.x < 16) ownrow[lane] = ((const f4_t *)lrow)[self_i * 16 + lane];   // the own horizon: 15 x (x, y, z, 0) + 4 zeros
    // own boxes (wave-uniform: scalar loads)
    const ConstF4 ob4 = (ConstF4)(unsigned long long)(bbox_nm + self_i * NBOX_NM);
    const f4_t o0 = ob4[0], o1 = ob4[1], o2 = ob4[2], o3 = ob4[3], o4 = ob4[4];
    const float obx[18] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w, o2.x, o2.y, o2.z, o2.w, o3.x, o3.y, o3.z, o3.w, o4.x, o4.y};
    __syncthreads();
    const int self_code = (g_local << 20) | cl;
    {
        const int sg = wave;
        // this segment's own box, inflated by the selection radius; the cells its neighbours' centres can lie in
        int c_lo[3], c_hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = sg == 0 ? obx[2 * a] : (sg == 1 ? obx[6 + 2 * a] : obx[12 + 2 * a]);
            const float hi = sg == 0 ? obx[2 * a + 1] : (sg == 1 ? obx[6 + 2 * a + 1] : obx[12 + 2 * a + 1]);
            const float infl = a < 2 ? R : Rz;
            const float reach = infl + __int_as_float(maxhalf[((size_t)scene * NSEG + sg) * 3 + a]);   // a neighbour's centre is at most its half extent from its box
            c_lo[a] = grid_coord(gg, a, lo - reach); c_hi[a] = grid_coord(gg, a, hi + reach);
        }
        // the own chord of this segment (z in the metric's scale) and the steps' largest distance from it, as grid_fill computes them
        f4_t oa0 = ownrow[SEG_STEPS * sg], oa1 = ownrow[SEG_STEPS * sg + SEG_STEPS - 1];
        oa0.z *= e1z; oa1.z *= e1z;
        const float odx = oa1.x - oa0.x, ody = oa1.y - oa0.y, odz = oa1.z - oa0.z;
        float odev2 = 0.f;
#pragma unroll
        for (int u = 1; u < SEG_STEPS - 1; ++u) {
            const f4_t o = ownrow[SEG_STEPS * sg + u];
            const float t = (float)u / (float)(SEG_STEPS - 1);
            const float dx = o.x - (oa0.x + t * odx), dy = o.y - (oa0.y + t * ody), dz = o.z * e1z - (oa0.z + t * odz);
            odev2 = fmaxf(odev2, dx * dx + dy * dy + dz * dz);
        }
        const float lim0 = sqrtf(thr2) + sqrtf(odev2) * 1.0001f + 2e-4f;   // selection radius + own deviation (+ slack for the fp32 arithmetic of the test)
        const size_t ss = (size_t)scene * NSEG + sg;
        const int *st = start + ss * (ncell + 1);
        const f4_t *en = ent + ss * nag, *en2 = ent + (size_t)S * NSEG * nag + ss * nag;
        int nst = 0;
        // the distance test over this segment's horizon steps on a full wave (or the rest) of staged survivors
        auto traj_test = [&](int n) {
            const bool have = lane < n;
            const int code = stage[have ? lane : 0];
            const int r = code >> 20, jc = code & 0xfffff;
            const f4_t *row = (const f4_t *)lrow + ((size_t)(r * S + scene) * C + jc) * 16 + SEG_STEPS * sg;
            f4_t v[SEG_STEPS];
#pragma unroll
            for (int u = 0; u < SEG_STEPS; ++u) v[u] = row[u];
            bool pass = false;
#pragma unroll
            for (int u = 0; u < SEG_STEPS; ++u) {
                const f4_t o = ownrow[SEG_STEPS * sg + u];
                const float dx = o.x - v[u].x, dy = o.y - v[u].y, dz = (o.z - v[u].z) * e1z;
                pass = pass || (dx * dx + dy * dy + dz * dz < thr2);
            }
            if (have && pass) {
                const int idx = r * C + jc;
                atomicOr(bits + (idx >> 5), 1u << (idx & 31));
            }
        };
        // The candidates: per (y, z) cell row one RUN of entries (the cells of a run along x are contiguous), ~50 entries each at N = 10^4.
        // The bounds of up to 64 runs are fetched at once (lanes = runs); the records of the next round are in flight during a round's test.
        const int ny = c_hi[1] - c_lo[1] + 1, nruns_all = ny * (c_hi[2] - c_lo[2] + 1);
        const UDiv div_ny((unsigned)ny);
        for (int run0 = 0; run0 < nruns_all; run0 += 64) {
            const int nruns = nruns_all - run0 < 64 ? nruns_all - run0 : 64;
            int rb = 0, re = 0;
            if (lane < nruns) {
                int qz, qy;
                div_ny.divmod((unsigned)(run0 + lane), qz, qy);
                const int row0 = ((c_lo[2] + qz) * gg.n[1] + c_lo[1] + qy) * gg.n[0];
                rb = st[row0 + c_lo[0]]; re = st[row0 + c_hi[0] + 1];
            }
            // The runs of this batch as ONE sequence of entries (a run holds ~50 entries at N = 10^4: a round of 64 lanes per run left a third of
            // the lanes idle and, worse, made every run a memory round trip of its own): lanes = the next 64 entries of the concatenation, each
            // lane finds its run in the prefix sums of the run lengths (six shuffles) -- half the rounds.
            const int len = re - rb;
            int pre = len;   // inclusive prefix of the run lengths
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(pre, off); if (lane >= off) pre += o; }
            const int tot = __builtin_amdgcn_readlane(pre, 63);
            const int basev = rb - (pre - len);   // entry index of sequence position v in run j: basev_j + v
            const int nround = (tot + 63) >> 6;
            struct Rec { f4_t a, b; bool hv; };
            // (the loads of a round are issued UNCONDITIONALLY, into registers of their own -- two record sets, the loop unrolled by two: a
            // fetch behind a branch, or a copy of the loaded registers at the loop's end, makes the compiler wait for the loads it has just
            // issued (s_waitcnt vmcnt(0)) and the prefetch is gone: that was the state of this loop until the instruction stream was read)
            auto fetch = [&](int k, Rec &rc) {
                const int vi = 64 * k + lane;
                rc.hv = vi < tot;
                const int vq = rc.hv ? vi : tot - 1;
                int j = 0;   // the first run whose inclusive prefix exceeds vq
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) { const int pm = __shfl(pre, j + step - 1); if (pm <= vq) j += step; }
                const size_t ei = (size_t)(__shfl(basev, j) + vq);
                rc.a = en[ei]; rc.b = en2[ei];
            };
            auto test = [&](const Rec &rc) {
                // closest approach of the two chords at equal time: r(t) = q + t e on [0, 1], against radius + both deviations.  (Any t gives
                // an upper bound of the minimum; the rounded t* is off by parts in 10^6 and |r(t)|^2 is flat there to second order.)
                const float qx = rc.a.y - oa0.x, qy = rc.a.z - oa0.y, qz = rc.a.w - oa0.z;
                const float ex = rc.b.x - odx, ey = rc.b.y - ody, ez = rc.b.z - odz;
                const float ee = ex * ex + ey * ey + ez * ez, qe = qx * ex + qy * ey + qz * ez, qq = qx * qx + qy * qy + qz * qz;
                const float tt = __builtin_fminf(__builtin_fmaxf(-qe * __builtin_amdgcn_rcpf(__builtin_fmaxf(ee, 1e-20f)), 0.f), 1.f);
                const float d2 = qq + tt * (qe + qe + tt * ee);   // |q + t e|^2 (rounding: parts in 10^7 of qq <= 60: the slack in lim0 covers it)
                const float lim = lim0 + rc.b.w;
                const int c0 = __float_as_int(rc.a.x);
                const bool surv = rc.hv && c0 != self_code && d2 <= lim * lim;
                const unsigned long long m = __ballot(surv);
                if (m) {
                    if (surv) stage[nst + lanes_below(m, lane)] = c0;
                    nst += __popcll(m);
                    LSYNC();
                    if (nst >= 64) {
                        traj_test(64);
                        LSYNC();
                        const int keep = lane < nst - 64 ? stage[64 + lane] : 0;
                        LSYNC();
                        if (lane < nst - 64) stage[lane] = keep;
                        nst -= 64;
                        LSYNC();
                    }
                }
            };
            if (nround > 0) {
                Rec ra, rb2;
                fetch(0, ra);
                for (int k = 0; k < nround; k += 2) {
                    fetch(k + 1 < nround ? k + 1 : nround - 1, rb2);   // (past the end: the last round once more, never tested)
                    test(ra);
                    if (k + 1 >= nround) break;
                    fetch(k + 2 < nround ? k + 2 : nround - 1, ra);
                    test(rb2);
                }
            }
        }
        if (nst > 0) traj_test(nst);
    }
    __syncthreads();
    if (wave != 0) return;
    // the list = the bitmap in order (increasing neighbour index), one contiguous run per agent
    int *out = list + (size_t)oid * cap;
    int total = 0;
    const UDiv div_c((unsigned)C);
    for (int w0 = 0; w0 < nwords; w0 += 64) {
        unsigned word = w0 + lane < nwords ? bits[w0 + lane] : 0u;
        const int mycnt = __popc(word);
        int pre = mycnt;   // inclusive prefix of the lanes' counts
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(pre, off); if (lane >= off) pre += o; }
        const int wave_total = __builtin_amdgcn_readlane(pre, 63);
        int pos = total + pre - mycnt;
        while (word) {
            const int b = __ffs((int)word) - 1;
            word &= word - 1u;
            const int idx = ((w0 + lane) << 5) + b;
            int r = 0, jc = idx;
            if (G > 1) div_c.divmod((unsigned)idx, r, jc);
            if (pos < cap) out[pos] = (r << 20) | jc;
            ++pos;
        }
        total += wave_total;
    }
    if (lane < NBR_PARTS) cnt_out[(size_t)oid * NBR_PARTS + lane] = lane == 0 ? (total > cap ? -1 : total) : 0;
}

// calibration of the HBM-side counters (profiles/: FETCH_SIZE is documented for 16-byte-per-lane streams only): a streaming read of a known
// number of bytes at 8 or 16 bytes per lane, coalesced, as this library's kernels read their tables and rows
template <typename T>
__global__ void read_probe_kernel(size_t n, const T *__restrict__ src, double *__restrict__ sink)
{
    double acc = 0.0;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const T v = src[t];
        acc += ((const double *)&v)[0];
    }
    if (acc == 1.2345e300) sink[0] = acc;   // (keeps the loads alive; never true for a zeroed buffer)
}

// Longest-processing-time-first launch order for the solve phase: agents are bucketed by the key the scan left in
// hdr[7] (row count for the slack-carrying variants, violated steps + tightness for the slack-free ones) and the solve
// kernel takes them heaviest first, so the long solves do not end up alone at the tail of the launch.  Pure scheduling:
// results do not depend on the order.  gridDim.x workgroups: workgroup b sorts the agents i = b (mod gridDim.x)
// (statistically identical slices) and writes its r-th heaviest agent to position r * gridDim.x + b, so the
// interleaved sequence is heaviest-first overall up to the differences between the slices.
// Order hint (round 4): when the context solved the same batch shape in its previous step, `prev_cost` holds every agent's work estimate of
// THAT solve (quarter microseconds, written by the solve kernel; 0: finished by the scan) -- in a closed loop the best predictor there is of
// this step's solve (an agent in a conflict stays in it for several steps), where the scan's key (row count) says little about the retry
// ladder or the size of the final working set: at N = 10^4 the launch took 1.6 times its work per wave slot and ended with agents of
// 100-450 us that the row count had put at the END of the queue.  key = max(previous cost / 4 us, scan key); the entry is reset to 0 for
// agents the scan finishes this step.
__global__ void order_kernel(int count, const int *__restrict__ hdr, int *__restrict__ order, int *__restrict__ live_bound, int *__restrict__ prev_cost, int use_hint)
{
    // (agents the scan already finished -- hdr[7] & 256, as hdr[4] & 16 -- sort behind everything else: bucket 256.  Slice b puts its live agents at
    // positions b, b + nb, ...: every position from nb * max_b(live agents of slice b) on holds a finished agent, and that bound
    // is what the solve queue runs to.)
    __shared__ int hist[257];
    __shared__ int offs[257];
    short *keys = (short *)dmpc_smem;   // dynamic LDS: the slice's keys, first pass -> second pass (ceil(count / gridDim.x) entries)
    const int nb = gridDim.x, b = blockIdx.x;
    for (int i = threadIdx.x; i < 257; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = b + nb * (int)threadIdx.x; i < count; i += nb * (int)blockDim.x) {
        const int h7 = hdr[(size_t)i * 8 + 7];
        int heavy = h7 & 255;
        if (prev_cost) {
            const int pc = prev_cost[i] >> 4;
            if (use_hint == 2 ? pc > 0 : (use_hint && pc > heavy)) heavy = pc > 255 ? 255 : pc;   // (2: the previous work alone -- in a replay of ONE step the perfect order: tools/gpu_order_oracle.py)
            if (h7 & 256) prev_cost[i] = 0;
        }
        const int key = (h7 & 256) ? 256 : 255 - heavy;   // bucket: heaviest first
        keys[(i - b) / nb] = (short)key;
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    {   // exclusive prefix of the 257 buckets: a scan inside each of the first four waves, their totals through LDS (one thread walking the buckets was 3 us of a 7 us kernel)
        __shared__ int wsum[4];
        const int t = (int)threadIdx.x, ln = t & 63, wv = t >> 6;
        int v = 0, inc = 0;
        if (t < 256) {
            v = hist[t]; inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (ln >= off) inc += o; }
            if (ln == 63) wsum[wv] = inc;
        }
        __syncthreads();
        if (t < 256) {
            int base = 0;
            for (int w = 0; w < wv; ++w) base += wsum[w];
            offs[t] = base + inc - v;
        }
        if (t == 0) {
            const int live = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            offs[256] = live;
            if (live_bound) atomicMax(live_bound, nb * live);
        }
    }
    __syncthreads();
    for (int i = b + nb * (int)threadIdx.x; i < count; i += nb * (int)blockDim.x) {
        const int key = keys[(i - b) / nb];
        order[(size_t)atomicAdd(&offs[key], 1) * nb + b] = i;
    }
}

// rows[S][N][3K] -> lT[G][S][3K][C], N = G*C
__global__ void table_from_rows_kernel(int S, int G, int C, const double *__restrict__ rows, double *__restrict__ lT)
{
    const size_t total = (size_t)S * G * C * N3;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        // t enumerates the destination (coalesced writes): [g][s][j][c]
        const int c = (int)(t % C);
        size_t u = t / C;
        const int j = (int)(u % N3); u /= N3;
        const int s = (int)(u % S);
        const int g = (int)(u / S);
        lT[t] = rows[((size_t)s * (G * C) + (size_t)g * C + c) * N3 + j];
    }
}

// mixed precision: the fp32 copy of a table the scan of the next step reads
__global__ void table_to_f32_kernel(size_t n, const double *__restrict__ src, float *__restrict__ dst)
{
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) dst[t] = (float)src[t];
}

// x <- first horizon column for solved agents (dmpc_soft_bound.m:132-134)
__global__ void advance_kernel(int count, const double *__restrict__ p, const double *__restrict__ v,
                               const double *__restrict__ a, const int *__restrict__ status, double *x_p, double *x_v, double *x_a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 3) return;
    const int ag = i / 3, d = i - 3 * ag;
    if (status[ag] & ST_SOLVED) {
        x_p[i] = p[(size_t)ag * N3 + d];
        x_v[i] = v[(size_t)ag * N3 + d];
        x_a[i] = a[(size_t)ag * N3 + d];
    }
}

// initDMPC.m:6-12: p(:,i) = po + t_i (pf - po)/10, v = a = 0 ; writes rows [count][3K]
__global__ void init_rows_kernel(int count, double h, const double *__restrict__ po, const double *__restrict__ pf,
                                 double *__restrict__ l_rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * N3) return;
    const int ag = i / N3, j = i - ag * N3;
    const int k = j / 3, d = j - 3 * k;
    const double t = (double)k * h;
    const double diff = pf[3 * ag + d] - po[3 * ag + d];
    l_rows[i] = po[3 * ag + d] + 1 * t * diff / 10;
}

// pk/vk/ak(:,k,n) = state after MPC step k   (dmpc_soft_bound.m:132-134); hist: [S][N][KT][3]
__global__ void record_kernel(int S, int N, int KT, int k, const double *__restrict__ x_p, const double *__restrict__ x_v,
                              const double *__restrict__ x_a, double *pk, double *vk, double *ak)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * N * 3) return;
    const int ag = i / 3, d = i - 3 * ag;
    const size_t o = ((size_t)ag * KT + k) * 3 + d;
    pk[o] = x_p[i]; vk[o] = x_v[i]; ak[o] = x_a[i];
}

// The three small kernels that follow a solve in a transition, in one launch (one block per scene; a 100-agent step is
// latency bound, and three launches cost more than their work): x <- first horizon column of the solved agents
// (dmpc_soft_bound.m:132-134), the history column k (pk/vk/ak(:,k,n)), then the scene verdict as scene_reduce_kernel.
__global__ void post_step_kernel(int N, int KT, int k, double tol, const double *__restrict__ p, const double *__restrict__ v,
                                 const double *__restrict__ a, const int *__restrict__ status, double *x_p, double *x_v, double *x_a,
                                 const double *__restrict__ pf, double *pk, double *vk, double *ak, int *flags, int *scene_done,
                                 const int *__restrict__ stopped)
{
    __shared__ double smax[256];
    __shared__ int sor[256];
    const int s = blockIdx.x;
    // a scene whose trial ended at an earlier step is left alone: its history columns stay zero, as the preallocated pk/vk/ak of
    // the reference do after its `break` (failure_rate.m:112-125)
    if (stopped && stopped[s]) return;
    double m = 0.0; int o = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const size_t ag = (size_t)s * N + i, b = ag * 3;
        const int st = status[ag];
        double xp[3], xv[3], xa[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (st & ST_SOLVED) { xp[d] = p[ag * N3 + d]; xv[d] = v[ag * N3 + d]; xa[d] = a[ag * N3 + d]; x_p[b + d] = xp[d]; x_v[b + d] = xv[d]; x_a[b + d] = xa[d]; }
            else { xp[d] = x_p[b + d]; xv[d] = x_v[b + d]; xa[d] = x_a[b + d]; }
            const size_t ho = (ag * KT + k) * 3 + d;
            pk[ho] = xp[d]; vk[ho] = xv[d]; ak[ho] = xa[d];
        }
        const double dx = xp[0] - pf[b], dy = xp[1] - pf[b + 1], dz = xp[2] - pf[b + 2];
        m = fmax(m, sqrt(dx * dx + dy * dy + dz * dz));
        o |= st;
    }
    smax[threadIdx.x] = m; sor[threadIdx.x] = o;
    __syncthreads();
    for (int w = blockDim.x / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + w]); sor[threadIdx.x] |= sor[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int reached = smax[0] < tol ? 1 : 0;
        flags[(size_t)s * 2] = reached; flags[(size_t)s * 2 + 1] = sor[0];
        if (scene_done && (reached || (sor[0] & ~ST_SOLVED))) scene_done[s] = 1;
    }
}

// one block per scene: flags[0] = ReachedGoal.m:3-11 (max_i ||p_i - pf_i|| < tol), flags[1] = OR of status bits
__global__ void scene_reduce_kernel(int N, double tol, const double *__restrict__ x_p, const double *__restrict__ pf,
                                    const int *__restrict__ status, int *flags, int *scene_done)
{
    __shared__ double smax[256];
    __shared__ int sor[256];
    const int s = blockIdx.x;
    double m = 0.0; int o = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const size_t b = ((size_t)s * N + i) * 3;
        const double dx = x_p[b] - pf[b], dy = x_p[b + 1] - pf[b + 1], dz = x_p[b + 2] - pf[b + 2];
        m = fmax(m, sqrt(dx * dx + dy * dy + dz * dz));
        o |= status[(size_t)s * N + i];
    }
    smax[threadIdx.x] = m; sor[threadIdx.x] = o;
    __syncthreads();
    for (int w = blockDim.x / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + w]); sor[threadIdx.x] |= sor[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int reached = smax[0] < tol ? 1 : 0;
        flags[(size_t)s * 2] = reached; flags[(size_t)s * 2 + 1] = sor[0];
        // the trial of this scene is over (failure_rate.m:112-125): later steps skip it on the device as well
        if (scene_done && (reached || (sor[0] & ~ST_SOLVED))) scene_done[s] = 1;
    }
}

}  // namespace dmpc
