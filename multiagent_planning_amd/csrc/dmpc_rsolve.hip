// dmpc_rsolve.hip -- the REDUCED solver of the one-step-row slack variants (solveSoftDMPCbound.m:43-160; round 6): included by
// dmpc_kernels.hip inside namespace dmpc, behind dmpc_solve.hip (whose DPP helpers it uses).
//
// Same method as dmpc_solve.hip -- the dual active-set iteration of Goldfarb and Idnani, same pivot rule, same retry ladder -- on another
// linear algebra.  solveSoftDMPCbound puts every collision row on ONE horizon step kc (:21-38), its Hessian is, per axis,
//   H1 = 2 (q lK lK' + s D'D + I)                       (:43-57,98; getDeltaMat.m:3-8; lK = the last row of the position map)
// = a constant tridiagonal matrix plus a rank-1 term, and its bounds |a| <= alim (:3-5) and slb <= eps <= 0 (:77-78) are bounds on single
// variables.  So
//   * an active acceleration bound is a FIXED VARIABLE, not a constraint: for any free set F the solve with H1_FF is a tridiagonal solve
//     (parallel cyclic reduction over the 16 lanes of the axis' DPP row) plus Sherman-Morrison -- registers only, no factor;
//   * a slack at one of its bounds is a fixed variable too: its row is then a HARD row (three at most are independent: they only see the
//     3-vector w_kc); a row with a free slack is a SOFT row and enters as the rank-1 penalty 2/sd^2 xi xi' on w_kc -- any number of
//     them cost nine wave reductions;
//   * what is left as "general constraints" -- hard rows, workspace walls, the entering constraint -- is a system of at most five
//     unknowns, solved from scratch in uniform registers with the entering constraint LAST (its pivot is the dependence test).
// Every equality-constrained QP of the iteration is solved FROM SCRATCH: nothing is updated, nothing drifts, no verification pass; a
// partial step interpolates multipliers (and the primal) between two such solutions.  No LDS beyond 96 doubles per wave for the output
// stage, so the launch is bound by registers, not by the 19 KB per wave of the inverse factor (dmpc_solve.hip: 1.75 waves per SIMD).
// CPU prototype of exactly this algorithm, validated against the oracle on 27 000 agent-steps: tools/proto/rqp_proto.c.
//
// Lane layout: component (axis x, step k) lives in lane 16 x + k (k < 15): the tridiagonal neighbours are row_shr:1 / row_shl:1, sums over
// an axis are sums over a DPP row; collision row j lives in lane j (at most 64 rows: more -> the general kernel).  Agents this kernel does
// not take (more than 64 rows, rows on several steps, more than two active walls, more than five hard constraints, an iteration cap) are
// flagged ST_QOVER and solved by the general kernel (dmpc_solve.hip) in the tier-2 launch: none in the 27 000 agent-steps of the prototype's
// campaign but for a third wall (1).

// #define RSOLVE_TRACE 1
template <int N> __device__ __forceinline__ double rshr(double v) { return dpp0_d<0x110 + N>(v); }   // lane i <- lane i-N of its row (0 off the row)
template <int N> __device__ __forceinline__ double rshl(double v) { return dpp0_d<0x100 + N>(v); }   // lane i <- lane i+N
template <int N> __device__ __forceinline__ double rror(double v) { return dpp0_d<0x120 + N>(v); }   // rotation inside the row
__device__ __forceinline__ double row_allsum(double v) { v += rror<8>(v); v += rror<4>(v); v += rror<2>(v); v += rror<1>(v); return v; }   // every lane: the sum over its row
__device__ __forceinline__ double row_prefix(double v) { v += rshr<1>(v); v += rshr<2>(v); v += rshr<4>(v); v += rshr<8>(v); return v; }   // inclusive prefix sum inside the row

enum { RE_BOUND = 0, RE_WALL = 1, RE_ROW = 2, RE_PIN0 = 3, RE_PINL = 4, RE_NONE = 5 };
enum { RB_IN = 1, RB_PIN0 = 2, RB_PINL = 4 };
constexpr int R_NH = 8;   // hard constraints of the small system, one per lane (hard rows + walls + the entering constraint)
constexpr int R_NW = 3;   // walls of the working set (a corner of the workspace)
constexpr int R_NE = R_NW + 1;
constexpr int REQP_MAX = 110;       // equality-constrained solves per ladder level before the agent is handed to the general solver
constexpr int RBLOCK_UNTIL = 48;    // ... and only among the first equality solves of a level: a level that runs longer goes on with the exact steps (a block move is not
                                    // monotone in the dual objective: an agent in a corner of the workspace cycled through walls and block moves with period 24)
constexpr int RBLOCK_MAX = 24;     // block moves of the bounds per ladder level (then the exact one-at-a-time steps only)
constexpr int RCERT_AFTER = 6;      // scans of a ladder level before the certificate looks at it
constexpr int RCERT_PLANES = 70;    // planes the wave's LDS holds for it: 64 rows + 6 box faces   // extras: the walls + an entering wall / bound

// the per-axis tridiagonal solver of a free set (PCR multipliers of the four strides), u = T3_FF^-1 lK_F and kap = 1 / (1 + 2 q lK_F'u)
struct RAx {
    double al[4], ga[4], binv, u, kap;
};

template <int S, int ST>
__device__ __forceinline__ void pcr_setup_step(RAx &A, double &a_, double &b_, double &c_, const int k)
{
    const double bm = rshr<S>(b_), cm = rshr<S>(c_), am = rshr<S>(a_);
    const double bp = rshl<S>(b_), ap = rshl<S>(a_), cp = rshl<S>(c_);
    const double al = (k >= S) ? -a_ * fast_rcp(bm) : 0.0;
    const double ga = (k + S < 16) ? -c_ * fast_rcp(bp) : 0.0;
    A.al[ST] = al; A.ga[ST] = ga;
    b_ = fma(al, cm, fma(ga, ap, b_));
    a_ = al * am; c_ = ga * cp;
}
// y = T3_FF^-1 r (r = 0 on the fixed lanes: their equation is the identity, their multipliers are zero)
__device__ __forceinline__ double pcr_apply(const RAx &A, double r)
{
    r = fma(A.al[0], rshr<1>(r), fma(A.ga[0], rshl<1>(r), r));
    r = fma(A.al[1], rshr<2>(r), fma(A.ga[1], rshl<2>(r), r));
    r = fma(A.al[2], rshr<4>(r), fma(A.ga[2], rshl<4>(r), r));
    r = fma(A.al[3], rshr<8>(r), fma(A.ga[3], rshl<8>(r), r));
    return r * A.binv;
}
// z = H1_FF^-1 nu (nu = 0 on the fixed lanes), Sherman-Morrison on the rank-1 term
__device__ __forceinline__ double rax_solve(const RAx &A, const double nu, const double lKl, const double q2)
{
    const double y = pcr_apply(A, nu);
    const double d = row_allsum(lKl * y);
    return fma(-(q2 * A.kap * d), A.u, y);
}
// (H1 v)_i, every lane of the row (v = 0 on the lanes that are no components)
__device__ __forceinline__ double rax_hmul(const double v, const double dg, const double e, const double lKl, const double q2)
{
    const double d = row_allsum(lKl * v);
    return fma(q2 * lKl, d, fma(e, rshr<1>(v) + rshl<1>(v), dg * v));
}

// B^-1 of the symmetric positive definite 3x3 matrix (b00 b01 b02; . b11 b12; . . b22) by cofactors (B = I + a positive semidefinite matrix)
struct Sym3 { double m00, m01, m02, m11, m12, m22; };
__device__ __forceinline__ Sym3 sym3_inv(const Sym3 &B)
{
    const double c00 = B.m11 * B.m22 - B.m12 * B.m12, c01 = B.m02 * B.m12 - B.m01 * B.m22, c02 = B.m01 * B.m12 - B.m02 * B.m11;
    const double det = B.m00 * c00 + B.m01 * c01 + B.m02 * c02;
    const double id = fast_rcp(det);
    Sym3 R;
    R.m00 = c00 * id; R.m01 = c01 * id; R.m02 = c02 * id;
    R.m11 = (B.m00 * B.m22 - B.m02 * B.m02) * id; R.m12 = (B.m01 * B.m02 - B.m00 * B.m12) * id; R.m22 = (B.m00 * B.m11 - B.m01 * B.m01) * id;
    return R;
}
__device__ __forceinline__ void sym3_mul(const Sym3 &M, const double *v, double *o)
{
    o[0] = M.m00 * v[0] + M.m01 * v[1] + M.m02 * v[2];
    o[1] = M.m01 * v[0] + M.m11 * v[1] + M.m12 * v[2];
    o[2] = M.m02 * v[0] + M.m12 * v[1] + M.m22 * v[2];
}

// a-space normal of wall `code` (component lane | sign bit 8): sg l_k on its axis
__device__ __forceinline__ double wall_normal(const int code, const bool comp, const int ax_l, const int k_l, const double h2)
{
    const int wl = code & 63, ka = wl & 15;
    return (comp && ax_l == (wl >> 4) && k_l <= ka) ? ((code & 256) ? 1.0 : -1.0) * h2 * ((double)(ka - k_l) + 0.5) : 0.0;
}

// one agent; the wave's 96 doubles of LDS (`smem`) serve the output stage only
//
// State of a level.  Component lanes (16 axis + step): a, fx (0 free, +-1 fixed at +-alim), mu (multiplier of the fixed bound; lanes 48 .. 48 + nw - 1,
// which are no components, hold the multipliers of the walls).  Row lanes: rfl (RB_IN the row is active, RB_PIN0 / RB_PINL its slack is fixed at 0 / slb) and
// ONE multiplier lam -- the slack and the multipliers of its bounds follow from it (stationarity in eps: 2 eps + st + sd lam + pi - rho = 0):
//   free slack  eps = -(sd lam + st) / 2;   pin at 0  pi = -st - sd lam;   pin at slb  rho = 2 slb + st + sd lam
// all linear in lam, so a partial step interpolates lam alone.
__device__ __forceinline__ void rsolve_body(const StepParams &P, const int lane, const int vb, unsigned char *smem, const bool want_ticket, int &ticket, bool &claimed)
{
#if defined(RSOLVE_TRACE)
    long long rph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long rph_last = __builtin_amdgcn_s_memtime();
#define RPH(i_) do { const long long t__ = __builtin_amdgcn_s_memtime(); rph[i_] += t__ - rph_last; rph_last = t__; } while (0)
#else
#define RPH(i_) do { } while (0)
#endif
#define RCLAIM_NEXT() do { if (want_ticket && !claimed) { claimed = true; if (lane == 0) ticket = atomicAdd(kernarg_params()->counter, 1); } } while (0)
    // (fields of the parameter block that are picked by a run-time index -- pmin / pmax by axis, the weights by cost case -- are read through the
    // kernel-argument segment: a select between fields of the by-value struct is compiled as a select of ADDRESSES and sends the whole block to scratch)
    const KargPtr Qk = kernarg_params();
#define PMAXQ(x_) ((x_) == 0 ? Qk->pmax[0] : ((x_) == 1 ? Qk->pmax[1] : Qk->pmax[2]))
#define PMINQ(x_) ((x_) == 0 ? Qk->pmin[0] : ((x_) == 1 ? Qk->pmin[1] : Qk->pmin[2]))
    const int nrmax = P.nrmax, var = P.variant;
    const int scene = vb / P.c_count, ci = vb - scene * P.c_count;
    const int cl = P.c_first + ci;
    const int gid = scene * P.c_count + ci;
    double *B = (double *)__builtin_assume_aligned(smem, 16);
    const size_t per = (size_t)nrmax * 7;
    const double *g_rows = P.rowbuf + (size_t)gid * per;
    const double *r_xi = g_rows, *r_b = g_rows + 3 * (size_t)nrmax, *r_sd = r_b + nrmax, *r_st = r_b + 2 * (size_t)nrmax, *r_slb = r_b + 3 * (size_t)nrmax;
    const int *r_kc = P.rowkc + (size_t)gid * nrmax;
    const int *hdr = P.hdr + (size_t)gid * 8;
    struct { int x, y, z, w; } h0, h1;
    h0.x = UNI(hdr[0]); h0.y = UNI(hdr[1]); h0.z = UNI(hdr[2]); h0.w = UNI(hdr[3]); h1.x = UNI(hdr[4]); h1.y = UNI(hdr[5]); h1.z = UNI(hdr[6]); h1.w = UNI(hdr[7]);
    double po[3], vo[3], ao[3], pf[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { po[d] = P.x_p[3 * gid + d]; vo[d] = P.x_v[3 * gid + d]; ao[d] = P.x_a[3 * gid + d]; pf[d] = P.pf[3 * gid + d]; }
    if (h1.x & 8) return;                              // agent of a scene that already stopped
    if (h1.x & 16) {                                   // finished by the scan (unconstrained exit)
        if (P.post_on) {
            const KargPtr Qp = kernarg_params();
            const int st_done = Qp->status[gid];
            double p1 = 0.0, v1 = 0.0, a1 = 0.0;
            if (lane < 3) { p1 = Qp->p_out[(size_t)gid * N3 + lane]; v1 = Qp->v_out[(size_t)gid * N3 + lane]; a1 = Qp->a_out[(size_t)gid * N3 + lane]; }
            post_step_part(Qp, lane, gid, scene, (st_done & ST_SOLVED) != 0, st_done, p1, v1, a1);
        }
        return;
    }
    const int nr = h0.x;
    int status = h0.w;
    const int nrows_built = h0.y, viol_k = h0.z;
    const bool violation = (h1.x & 1) != 0, rows_exist = h1.y != 0;
    const bool cppv = (var == VAR_CPP || var == VAR_CPP2);

    // ---------------------------------------------------------------- rows: lane = row
    const bool rv = lane < nr;
    const int ri = rv ? lane : 0;
    double xi0 = r_xi[3 * ri], xi1 = r_xi[3 * ri + 1], xi2 = r_xi[3 * ri + 2];
    double rb = r_b[ri], rsd = r_sd[ri];
    const double rst_l = r_st[ri], rslb_l = r_slb[ri];
    const int rkc = r_kc[ri];
    if (!rv) { xi0 = xi1 = xi2 = 0.0; rb = 0.0; rsd = 1.0; }
    const int kc = (nr > 0) ? UNI(rkc) : 0;
    // the slacks' linear cost and lower bound are the same for every row of these variants (solveSoftDMPCbound.m:78,82): wave-uniform
    double st = (nr > 0) ? readlane_d(rst_l, 0) : 0.0, slb = (nr > 0) ? readlane_d(rslb_l, 0) : 0.0;
    bool giveup = nr > 64 || __ballot(rv && (rkc != kc || rst_l != st || rslb_l != slb)) != 0ull;
    int why = giveup ? (nr > 64 ? 1 : 2) : 0;   // development: why the agent goes to the general solver
    const double risd = fast_rcp(rsd);                                                  // 1 / sd
    const float rwt = 4.f * __builtin_amdgcn_rsqf((float)(xi0 * xi0 + xi1 * xi1 + xi2 * xi2));   // pivot weight of the row ("rows first", dmpc_solve.hip)

    // ---------------------------------------------------------------- cost case, component constants: lane = 16 axis + step
    const int ccase = UNI(cost_case(var, po[0] - pf[0], po[1] - pf[1], po[2] - pf[2], rows_exist));
    const double qw = ccase == 0 ? Qk->Qfar : (ccase == 1 ? Qk->Qnear : Qk->Q1);
    const double sw = ccase == 2 ? Qk->S1 : Qk->Sfree;
    const double q2 = 2.0 * qw, e_off = -2.0 * sw;
    const int ax_l = lane >> 4, k_l = lane & 15;
    const bool comp = ax_l < 3 && k_l < K;
    const double h2 = P.h * P.h;
    const double lKl = comp ? h2 * ((double)(K - 1 - k_l) + 0.5) : 0.0;                  // lK(k)
    const double lkc = (comp && k_l <= kc) ? h2 * ((double)(kc - k_l) + 0.5) : 0.0;      // l_kc(k)
    const double dg = comp ? (k_l < K - 1 ? 4.0 * sw + 2.0 : 2.0 * sw + 2.0) : 0.0;
    const double *Gt = P.tables + (size_t)ccase * TAB_CASE_DOUBLES;
    const int kt = comp ? k_l : 0;
    double f_l = 0.0, pbase = 0.0, a_unc = 0.0, whi_l = 0.0, wlo_l = 0.0;
    {
        const double gax = comp ? goal_gap(sel3(pf, ax_l), sel3(po, ax_l), sel3(vo, ax_l), P.h) : 0.0;
        const double ao_l = comp ? sel3(ao, ax_l) : 0.0;
        f_l = comp ? (-q2 * lKl * gax - (k_l == 0 ? 2.0 * sw * ao_l : 0.0)) : 0.0;
        pbase = comp ? sel3(po, ax_l) + (double)(k_l + 1) * P.h * sel3(vo, ax_l) : 0.0;   // A_initp(k,:) [po; vo]: the walls are pmin - pbase <= w <= pmax - pbase
        whi_l = comp ? PMAXQ(ax_l) - pbase : INFINITY; wlo_l = comp ? PMINQ(ax_l) - pbase : -INFINITY;
        // unconstrained minimiser from the Gram tables, exactly as the scan's unconstrained exit and the general solver form it
        a_unc = comp ? unc_entry(qw, sw, gax, ao_l, Gt[kt * 30 + 15 + (K - 1)], Gt[kt * 30]) : 0.0;
    }
    // the scales of the bounds and walls, H1^-1(k,k) and (L H1^-1 L')(k,k), are read where an entering bound / wall needs one: a SCALAR load from the table
    // (constant address space: the index is wave-uniform) issued a section ahead of its use -- kept per lane they were four registers of a kernel that spills
    typedef const double __attribute__((address_space(4))) *ConstD;
    const ConstD Gts = (ConstD)(unsigned long long)Gt;
    const double sc_row = Gt[(15 + kc) * 31];   // n'H^-1 n of the UNREDUCED Hessian: the scale of the dependence test (dmpc_solve.hip: delta <= 1e-13 s_pp)

    // the ladder certificate (ladder_level_infeasible, dmpc_kernels.hip) wants the walls of component (k, axis) in lane 3 k + axis
    // (formed at the two calls, not kept: four registers for the whole solve)
#define RWALLS_STACKED(v_) __shfl((v_), lane < N3 ? 16 * (lane % 3) + lane / 3 : 63)
    const bool ladder = (var == VAR_BOUND || var == VAR_BOUND2 || cppv);
    const int max_tries = P.max_tries > 0 ? P.max_tries : (cppv ? 21 : 30);
    const double tol = 1e-10;
    int tries = h1.z, iters_total = 0, maxq = 0, qfinal = 0, cost = 0;
    double lev_f = 1.0;   // the rows' slack bound and penalty carry this factor (a power of two) on the current ladder level
    bool solved = false;
    double a = 0.0, lam = 0.0;
    int fx = 0, rfl = 0, nw = 0;

    if (status & ST_INFEAS) tries = 1;
    if (tries > 0 && !(status & ST_INFEAS)) {
        if (tries >= max_tries) { status |= ST_INFEAS; tries = max_tries; }
        else { const double f = ldexp(1.0, tries); slb *= f; st *= f; lev_f = f; }
    }
    if (!(status & (ST_COLL | ST_CAPACITY | ST_INFEAS)) && !giveup) {
        bool cert_known = false;   // the level about to start has passed the certificate
        while (tries < max_tries) {
            tries++;
            int rc = 0;   // 0 solved, 1 infeasible, 2 give up
            int lev_skip = 0;
            // ---- state of the level
            fx = 0; rfl = rv ? RB_PIN0 : 0; nw = 0;
            a = a_unc; lam = 0.0;
            double mu = 0.0;
            int wcode[R_NW] = {0, 0, 0};    // walls of the working set: lane of the component | sign bit 8
            RAx A;
            double a0 = 0.0, Ykc = 0.0;
            double w03[3] = {0, 0, 0}, sg3[3] = {0, 0, 0}, isg3[3] = {0, 0, 0};
            bool fdirty = true;
            // the soft rows' penalty, kept up to date one row at a time: M_s = sum 2/sd^2 xi xi', m_s = sum (2 b/sd^2 + st/sd) xi
            Sym3 Ms; Ms.m00 = Ms.m01 = Ms.m02 = Ms.m11 = Ms.m12 = Ms.m22 = 0.0;
            double msv[3] = {0.0, 0.0, 0.0};
            unsigned long long softm_prev = 0ull;
            // (gathered for every solve: kept across solves behind a key they were eleven registers of the loop's state -- and the key was wrong once, section 2 of DESIGN.md)
            int ent = RE_NONE, eidx = 0, esg = 0;   // entering constraint: type, lane of the component / row, sign
            int phase = 0;                  // 1: crash (free the negative multipliers), 2: iteration, 3: violation scan
            int inner = 0, iters = 0, zero_steps = 0, leqp = 0;
            // Block moves of the acceleration bounds.  An equality-constrained QP costs the same whatever changed since the last one, and the bounds flip
            // in blocks: a row that enters pushes a dozen of the fixed accelerations off their bounds, which the ratio test frees one partial step at
            // a time, and a dozen others are then violated and enter one full step at a time (an agent of 79 steps: 60 of them such).  So: (a) when the
            // scan's choice is a bound, EVERY violated bound is fixed at once; (b) when a bound's multiplier blocks the entering constraint, the entering
            // constraint joins and every bound whose multiplier would turn negative is freed at once -- then, as in the crash start, the bounds with
            // negative multipliers are freed until none is left.  The result is accepted if every multiplier of the working set is non-negative: it is
            // then a state of the dual method like any other (the minimiser of its working set, dual feasible).  If a row's, pin's or wall's
            // multiplier comes out negative the move is taken back and the exact step is made.  A budget per level keeps the method finite.
            bool blk = false, noblock = false;
#define RBLOCK_UNDO() do { fx = ((fxhi_s >> lane) & 1ull) ? 1 : (((fxlo_s >> lane) & 1ull) ? -1 : 0); rfl = rfl_s; \
                           blk = false; noblock = true; fdirty = true; phase = sphase; inner = 0; } while (0)
            int nblock = 0, sphase = 0, rfl_s = 0;
            unsigned long long fxhi_s = 0ull, fxlo_s = 0ull;
            bool cert_done = cert_known;
            cert_known = false;
            {   // crash start: every bound violated at the unconstrained minimiser is fixed
                const bool viol = comp && fabs(a_unc) - P.alim > tol;
                if (__ballot(viol) != 0ull) { fx = viol ? (a_unc > 0.0 ? 1 : -1) : 0; phase = 1; }
                else phase = 3;             // straight to the first violation scan: the unconstrained minimiser is the state
            }
            for (;;) {
                // =========================================================== violation scan (state: the minimiser of the working set)
#ifdef RSOLVE_MARK
                asm volatile("; @@R SCAN" ::: "memory");
#endif
                RPH(11);
                if (phase == 3) {
                    // positions w = Lambda a per axis by two prefix sums: w_k = h^2 ((k + 1/2) S0_k - S1_k)
                    const double s0 = row_prefix(comp ? a : 0.0), s1 = row_prefix(comp ? (double)k_l * a : 0.0);
                    const double w = h2 * fma((double)k_l + 0.5, s0, -s1);
                    const double wk0 = readlane_d(w, kc), wk1 = readlane_d(w, 16 + kc), wk2 = readlane_d(w, 32 + kc);
                    float bests = 0.f; int bestc = -1;
#define RCAND(v_, w_, code_) do { const double v__ = (v_); const float s__ = (float)v__ * (w_); if (v__ > tol && s__ > bests) { bests = s__; bestc = (code_); } } while (0)
                    if (comp) {
                        if (fx == 0) RCAND(fabs(a) - P.alim, 1.f, (RE_BOUND << 16) | (a > 0.0 ? 256 : 0) | lane);
                        const bool inw = (nw > 0 && (wcode[0] & 63) == lane) || (nw > 1 && (wcode[1] & 63) == lane) || (nw > 2 && (wcode[2] & 63) == lane);
                        const double c2 = w - whi_l, c3 = wlo_l - w;
                        if (!inw) RCAND(fmax(c2, c3), 1.f, (RE_WALL << 16) | (c2 > c3 ? 256 : 0) | lane);
                    }
                    if (rv) {
                        // (a lane is a component AND a row: the row's candidates compete with the component's through the same best-of)
                        if (!(rfl & RB_IN)) RCAND(-(xi0 * wk0 + xi1 * wk1 + xi2 * wk2) - rb, rwt, (RE_ROW << 16) | lane);
                        else if (!(rfl & (RB_PIN0 | RB_PINL))) {
                            const double eps = -0.5 * fma(rsd, lam, st);
                            RCAND(eps, 1.4142135f, (RE_PIN0 << 16) | lane);
                            RCAND(slb - eps, 1.4142135f, (RE_PINL << 16) | lane);
                        }
                    }
#undef RCAND
                    const float smax = wave_max_f(bests);
                    const unsigned long long wm = __ballot(bestc >= 0 && bests == smax);
                    if (wm == 0ull) { RCLAIM_NEXT(); rc = 0; break; }   // optimal
                    if (++iters > P.iter_cap || iters > 400) { rc = 2; why = 3; break; }
                    // the level is looked at by the ladder certificate once (3-variable polytope emptiness over the rows of the step with every slack at its
                    // bound): an infeasible level costs the dual method tens of steps to prove
                    if (ladder && violation && !cert_done && iters > RCERT_AFTER) {
                        cert_done = true;
                        cost += 100;
                        if (uni_b(ladder_level_infeasible(r_xi, r_b, r_sd, r_slb, r_kc, nr, B, P.h, P.alim, lev_f, RWALLS_STACKED(whi_l), RWALLS_STACKED(wlo_l), lane, RCERT_PLANES))) { rc = 1; break; }
                    }
                    const int pcode = readlane_i(bestc, __ffsll((long long)wm) - 1);
                    ent = pcode >> 16; eidx = pcode & 63; esg = (pcode & 256) ? 1 : -1;
                    if (ent == RE_WALL && nw >= R_NW) { rc = 2; why = 5; break; }
                    if (ent == RE_BOUND && !noblock && nblock < RBLOCK_MAX && leqp < RBLOCK_UNTIL) {
                        const bool vb_ = comp && fx == 0 && fabs(a) - P.alim > tol;
                        if (__popcll(__ballot(vb_)) >= 2) {
                            fxhi_s = __ballot(comp && fx > 0); fxlo_s = __ballot(comp && fx < 0); rfl_s = rfl; sphase = 3;
                            if (vb_) fx = a > 0.0 ? 1 : -1;
                            blk = true; ++nblock; fdirty = true; phase = 1; inner = 0;
                        }
                    }
                    if (phase == 3) {
                    // rows and pins join the working set at once (they stay "entering": their own multiplier does not block)
                    if (lane == eidx) {
                        if (ent == RE_ROW) rfl |= RB_IN;
                        if (ent == RE_PIN0) rfl |= RB_PIN0;
                        if (ent == RE_PINL) rfl |= RB_PINL;
                    }
                    inner = 0;
                    phase = 2;
                    }
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R EQP_START" ::: "memory");
#endif
                RPH(0);
                if (phase == 2 && ++inner > 200) { rc = 2; why = 4; break; }
                if (++leqp > (P.rsolve_cap > 0 ? P.rsolve_cap : REQP_MAX)) { rc = 2; why = 3; break; }   // (the longest level of 90 000 agent-steps of the 10^4-agent scene took 80: what runs on is a degenerate cycle)
                cost += 6;
                iters_total++;
                // =========================================================== the equality-constrained QP of the working set (+ entering constraint)
                const bool fixd = !comp || fx != 0;
#ifdef RSOLVE_MARK
                asm volatile("; @@R FDIRTY" ::: "memory");
#endif
                RPH(1);
                if (fdirty) {
                    fdirty = false;
                    cost += 2;
                    // PCR of T3_FF (fixed lanes and the lanes that are no components: identity equations)
                    const int fl = fixd ? 1 : 0;
                    const int fm = __builtin_amdgcn_update_dpp(1, fl, 0x111, 0xf, 0xf, false), fp = __builtin_amdgcn_update_dpp(1, fl, 0x101, 0xf, 0xf, false);   // neighbours' flags (1 off the row)
                    double b_ = fixd ? 1.0 : dg;
                    double a_ = (!fixd && !fm) ? e_off : 0.0, c_ = (!fixd && !fp) ? e_off : 0.0;
                    pcr_setup_step<1, 0>(A, a_, b_, c_, k_l);
                    pcr_setup_step<2, 1>(A, a_, b_, c_, k_l);
                    pcr_setup_step<4, 2>(A, a_, b_, c_, k_l);
                    pcr_setup_step<8, 3>(A, a_, b_, c_, k_l);
                    A.binv = fast_rcp(b_);
                    A.u = pcr_apply(A, fixd ? 0.0 : lKl);
                    A.kap = fast_rcp(fma(q2, row_allsum(fixd ? 0.0 : lKl * A.u), 1.0));
                    // a0: minimiser over the free components with the fixed ones at their bounds
                    const double ab = comp ? (double)fx * P.alim : 0.0;
                    const double hb = rax_hmul(ab, dg, e_off, lKl, q2);
                    a0 = rax_solve(A, fixd ? 0.0 : -f_l - hb, lKl, q2);
                    if (fixd) a0 = ab;
                    Ykc = rax_solve(A, fixd ? 0.0 : lkc, lKl, q2);
                    const double gl = row_allsum(lkc * Ykc), w0l = row_allsum(lkc * a0);
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        const double g = readlane_d(gl, 16 * x);
                        w03[x] = readlane_d(w0l, 16 * x);
                        const bool pos = g > 1e-300;
                        const double ir_ = fast_rsq(pos ? g : 1.0);
                        sg3[x] = pos ? g * ir_ : 0.0;
                        isg3[x] = pos ? ir_ : 0.0;
                    }
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R SOFT" ::: "memory");
#endif
                RPH(2);
                // ---- soft rows: M_s = sum 2/sd^2 xi xi', q_s = M_s w0 + sum (2 b/sd^2 + st/sd) xi;  B = I + sqrt(G) M_s sqrt(G)
                const bool r_in = rv && (rfl & RB_IN), r_hard = r_in && (rfl & (RB_PIN0 | RB_PINL)), r_soft = r_in && !r_hard;
                Sym3 Bi; Bi.m00 = 1.0; Bi.m11 = 1.0; Bi.m22 = 1.0; Bi.m01 = 0.0; Bi.m02 = 0.0; Bi.m12 = 0.0;
                double qt[3] = {0.0, 0.0, 0.0};
                {
                    const unsigned long long softm = __ballot(r_soft);
                    unsigned long long df = softm ^ softm_prev;
                    softm_prev = softm;
                    while (df != 0ull) {   // (a row at a time: one or none per step)
                        const int j = __ffsll((long long)df) - 1; df &= df - 1ull;
                        const double sgn = ((softm >> j) & 1ull) ? 1.0 : -1.0;
                        const double x0 = readlane_d(xi0, j), x1 = readlane_d(xi1, j), x2 = readlane_d(xi2, j), bj = readlane_d(rb, j), isj = readlane_d(risd, j);
                        const double al = sgn * 2.0 * isj * isj, be = sgn * (2.0 * bj * isj + st) * isj;
                        Ms.m00 = fma(al * x0, x0, Ms.m00); Ms.m01 = fma(al * x0, x1, Ms.m01); Ms.m02 = fma(al * x0, x2, Ms.m02);
                        Ms.m11 = fma(al * x1, x1, Ms.m11); Ms.m12 = fma(al * x1, x2, Ms.m12); Ms.m22 = fma(al * x2, x2, Ms.m22);
                        msv[0] = fma(be, x0, msv[0]); msv[1] = fma(be, x1, msv[1]); msv[2] = fma(be, x2, msv[2]);
                    }
                    if (softm == 0ull) { Ms.m00 = Ms.m01 = Ms.m02 = Ms.m11 = Ms.m12 = Ms.m22 = 0.0; msv[0] = msv[1] = msv[2] = 0.0; }   // (nothing left of the sums' round-off)
                    else {
                        cost += 1;
                        double qs[3];
                        sym3_mul(Ms, w03, qs);
                        qs[0] += msv[0]; qs[1] += msv[1]; qs[2] += msv[2];
                        Sym3 Bm;
                        Bm.m00 = fma(sg3[0] * sg3[0], Ms.m00, 1.0); Bm.m11 = fma(sg3[1] * sg3[1], Ms.m11, 1.0); Bm.m22 = fma(sg3[2] * sg3[2], Ms.m22, 1.0);
                        Bm.m01 = sg3[0] * sg3[1] * Ms.m01; Bm.m02 = sg3[0] * sg3[2] * Ms.m02; Bm.m12 = sg3[1] * sg3[2] * Ms.m12;
                        Bi = sym3_inv(Bm);
                        qt[0] = sg3[0] * qs[0]; qt[1] = sg3[1] * qs[1]; qt[2] = sg3[2] * qs[2];
                    }
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R HARDLIST" ::: "memory");
#endif
                RPH(3);
                // ---- the hard list, ONE CONSTRAINT PER LANE (lanes 0 .. nh-1): the hard rows of the working set in row order, the entering row
                // (while it is hard) last among them, then the extras (walls of the working set, an entering wall / bound).  Lane c holds
                // yt_c = its normal in the scaled w-space, its right-hand sides and row c of the small matrix S = Y' B^-1 Y + Om; the elimination
                // broadcasts one pivot row at a time (Gauss-Jordan on a positive definite matrix, the entering constraint LAST: its pivot is
                // delta = n_p' P n_p, the dependence test of the dual method).
                const bool ent_extra = (phase == 2) && (ent == RE_BOUND || ent == RE_WALL);
                const int ne = nw + (ent_extra ? 1 : 0);
                int erow = (phase == 2 && ent >= RE_ROW && ent <= RE_PINL) ? eidx : -1;
                if (erow >= 0 && !((readlane_i(rfl, erow) & RB_IN) && (readlane_i(rfl, erow) & (RB_PIN0 | RB_PINL)))) erow = -1;   // soft by now (its pin gave way): no bordered constraint
                unsigned long long hm = __ballot(r_hard);
                if (erow >= 0) hm &= ~(1ull << erow);
                const int nhr0 = __popcll(hm);
                const int nhr = nhr0 + (erow >= 0 ? 1 : 0);
                const int nh = nhr + ne;
                const int xb = nhr0;                      // the extras' lanes: xb .. xb + ne - 1 (an entering extra is the last of them = lane nh - 1)
                const int erl = erow >= 0 ? nh - 1 : -1;   // the entering row's lane: the last of the list, behind the walls
                if (nh > R_NH) { if (blk) { RBLOCK_UNDO(); continue; } rc = 2; why = 6; break; }
                const bool has_p = (phase == 2) && (ent_extra || erow >= 0);
                double hy0 = 0.0, hy1 = 0.0, hy2 = 0.0, hrho = 0.0, hd = 0.0, hsc = 1.0;
                double gx0 = 0.0, gx1 = 0.0, gx2 = 0.0, gb = 0.0, gsd = 1.0;
                int gfl = 0;
                if (nhr > 0) {
                    // (the key holds the entering row's LANE too: a wall that leaves during the row's inner iteration moves it down one lane with the same rows -- a
                    // gather keyed on the rows alone left row 0's data there: scene 452 of campaign seed 602, the one wrong answer of 2.4 M agent-steps)
                    {
                        int src = 0;
                        unsigned long long m = hm;
#pragma unroll
                        for (int c = 0; c < R_NH; ++c) {
                            if (c >= nhr0) break;
                            const int j = __ffsll((long long)m) - 1; m &= m - 1ull;
                            if (lane == c) src = j;
                        }
                        if (lane == erl) src = erow;
                        gx0 = __shfl(xi0, src); gx1 = __shfl(xi1, src); gx2 = __shfl(xi2, src); gb = __shfl(rb, src); gsd = __shfl(rsd, src);
                        gfl = __shfl(rfl, src);
                    }
                    if (lane < nhr0 || lane == erl) {
                        hy0 = -sg3[0] * gx0; hy1 = -sg3[1] * gx1; hy2 = -sg3[2] * gx2;
                        hd = gb - ((gfl & RB_PINL) ? gsd * slb : 0.0);
                        hrho = hd + (gx0 * w03[0] + gx1 * w03[1] + gx2 * w03[2]);
                        hsc = sc_row * (gx0 * gx0 + gx1 * gx1 + gx2 * gx2);
                    }
                }
                double omr[R_NE] = {0.0, 0.0, 0.0, 0.0};   // lane xb + i: row i of Om = G0 - yt yt' over the extras
                double Yp = 0.0;                             // H~ n of the entering bound (the common extra): kept for the update of a
                if (ne > 0) {
                    cost += 2;
                    if (ne == 1 && ent_extra && ent == RE_BOUND) {   // the entering bound alone: unit normal, every sum is a single entry
                        const double sgd = (double)esg;
                        Yp = rax_solve(A, (lane == eidx) ? sgd : 0.0, lKl, q2);
                        const int x = eidx >> 4;
                        const double t = sgd * readlane_d(Ykc, eidx) * (x == 0 ? isg3[0] : (x == 1 ? isg3[1] : isg3[2]));   // yt on its axis
                        const double g0 = sgd * readlane_d(Yp, eidx), u0 = sgd * readlane_d(a0, eidx);
                        if (lane == xb) { hy0 = x == 0 ? t : 0.0; hy1 = x == 1 ? t : 0.0; hy2 = x == 2 ? t : 0.0; hd = P.alim; hrho = P.alim - u0; hsc = Gts[(eidx & 15) * 31]; omr[0] = g0 - t * t; }
                    } else {
                        // walls (and maybe an entering wall / bound), one at a time: normal and H~ n are formed here and again for the update of a -- nothing of
                        // them is kept; an extra's normal lives on ONE axis, so yt_i is a scalar on that axis and G0_ij = n_j' H~ n_i needs extra i's H~ n only
                        double yts[R_NE] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int i = 0; i < R_NE; ++i) {
                            if (i >= ne) continue;
                            const bool is_b = ent_extra && i == nw && ent == RE_BOUND;
                            const int code = i < nw ? wcode[i < R_NW ? i : 0] : (eidx | (esg > 0 ? 256 : 0));
                            const int wl = code & 63, xa = wl >> 4;
                            const double nbi = is_b ? ((lane == eidx) ? (double)esg : 0.0) : wall_normal(code, comp, ax_l, k_l, h2);
                            const double Yi = rax_solve(A, fixd ? 0.0 : nbi, lKl, q2);
                            const double tl = row_allsum(lkc * Yi), ul = row_allsum(nbi * a0);
                            yts[i] = readlane_d(tl, wl & 48) * (xa == 0 ? isg3[0] : (xa == 1 ? isg3[1] : isg3[2]));
                            const double u0 = readlane_d(ul, wl & 48);
                            const double dd = is_b ? P.alim : ((code & 256) ? readlane_d(whi_l, wl) : -readlane_d(wlo_l, wl));
                            const double sc = is_b ? Gts[(wl & 15) * 31] : Gts[(15 + (wl & 15)) * 31];
                            if (lane == xb + i) { hy0 = xa == 0 ? yts[i] : 0.0; hy1 = xa == 1 ? yts[i] : 0.0; hy2 = xa == 2 ? yts[i] : 0.0; hd = dd; hrho = dd - u0; hsc = sc; }
#pragma unroll
                            for (int j = 0; j <= i; ++j) {
                                const bool jb = ent_extra && j == nw && ent == RE_BOUND;
                                const int cj = j < nw ? wcode[j < R_NW ? j : 0] : (eidx | (esg > 0 ? 256 : 0));
                                const double nbj = (j == i) ? nbi : (jb ? ((lane == eidx) ? (double)esg : 0.0) : wall_normal(cj, comp, ax_l, k_l, h2));
                                const bool same = ((cj & 63) >> 4) == xa;
                                const double g0 = readlane_d(row_allsum(nbj * Yi), wl & 48);   // (0 when the two are on different axes)
                                const double o = g0 - (same ? yts[i] * yts[j] : 0.0);
                                if (lane == xb + i) omr[j] = o;
                                if (lane == xb + j) omr[i] = o;
                            }
                        }
                    }
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R SROWS" ::: "memory");
#endif
                RPH(5);
                // row c of S and the right-hand side in lane c
                double Sr[R_NH], rh, hlam = 0.0, csave = 0.0;
                double zeta[3] = {0.0, 0.0, 0.0};
                int sing = 0;
                const int ie = lane - xb;   // (an extra's index; the entering row's lane xb + ne is none)
                {
                    const double hv[3] = {hy0, hy1, hy2};
                    double by[3], bq[3];
                    sym3_mul(Bi, hv, by);
                    sym3_mul(Bi, qt, bq);
                    rh = (lane < nh) ? -hrho - (hy0 * bq[0] + hy1 * bq[1] + hy2 * bq[2]) : 0.0;
#pragma unroll
                    for (int e = 0; e < R_NH; ++e) {
                        Sr[e] = 0.0;
                        if (e >= nh) continue;
                        const double b0 = readlane_d(by[0], e), b1 = readlane_d(by[1], e), b2 = readlane_d(by[2], e);
                        double t = hy0 * b0 + hy1 * b1 + hy2 * b2;
                        if (e >= xb && e < xb + ne && ie >= 0 && ie < ne) t += (e - xb == 0) ? omr[0] : ((e - xb == 1) ? omr[1] : ((e - xb == 2) ? omr[2] : omr[3]));
                        Sr[e] = (lane < nh) ? t : 0.0;
                    }
#ifdef RSOLVE_MARK
                asm volatile("; @@R GJ" ::: "memory");
#endif
                RPH(6);
                    // Gauss-Jordan, pivots in list order; the entering constraint's pivot decides dependence
                    bool bad = false;
                    double sd0 = Sr[0];   // my diagonal entry before the elimination: n_c' P n_c
#pragma unroll
                    for (int e = 1; e < R_NH; ++e) sd0 = (lane == e) ? Sr[e] : sd0;
#pragma unroll
                    for (int k = 0; k < R_NH; ++k) {
                        if (k >= nh) continue;
                        const double piv = readlane_d(Sr[k], k), psc = readlane_d(hsc, k);
                        if (k == nh - 1 && has_p) csave = Sr[k];   // (the entering constraint's column in the eliminated rows, before its own pivot)
                        // (the entering constraint also counts as dependent when the others take all but 1e-10 of its own norm in the reduced metric: fixing the
                        // last free acceleration a hard row can feel leaves a pivot of 1e-12 that passes the absolute test -- and a working set whose next
                        // elimination fails on the row)
                        const bool last = k == nh - 1 && has_p;
                        if (!(piv > (blk ? 1e-9 : 1e-13) * psc) || (last && !(piv > 1e-10 * readlane_d(sd0, k)))) {   // (a block move must leave a well-conditioned working set: it fixes bounds without the dependence test of an entering constraint)
                            if (k == nh - 1 && has_p) sing = 1; else bad = true;
                            continue;
                        }
                        const double f = (lane == k) ? 0.0 : Sr[k] * fast_rcp(piv);
                        rh = fma(-f, readlane_d(rh, k), rh);
#pragma unroll
                        for (int e = 0; e < R_NH; ++e) {
                            if (e <= k || e >= nh) continue;
                            Sr[e] = fma(-f, readlane_d(Sr[e], k), Sr[e]);
                        }
                        if (lane != k) Sr[k] = 0.0;
                    }
                    if (uni_b(bad)) { if (blk) { RBLOCK_UNDO(); continue; } rc = 2; why = 7; break; }   // (a block move that made the working set dependent is taken back)
                    sing = UNI(sing);
#ifdef RSOLVE_MARK
                asm volatile("; @@R SOLVE_TAIL" ::: "memory");
#endif
                RPH(7);
                    double dgn = Sr[0];   // my diagonal entry
#pragma unroll
                    for (int e = 1; e < R_NH; ++e) dgn = (lane == e) ? Sr[e] : dgn;
                    const double idg = (lane < nh && dgn > 0.0) ? fast_rcp(dgn) : 0.0;
                    if (!sing) {
                        hlam = rh * idg;
                        if (nh > 0) {
                            // zeta = -B^-1 (qt + sum_c yt_c lam_c): sums over the eight lanes that can hold a constraint
                            double p0 = hy0 * hlam, p1 = hy1 * hlam, p2 = hy2 * hlam;
                            p0 += rshr<1>(p0); p0 += rshr<2>(p0); p0 += rshr<4>(p0);
                            p1 += rshr<1>(p1); p1 += rshr<2>(p1); p1 += rshr<4>(p1);
                            p2 += rshr<1>(p2); p2 += rshr<2>(p2); p2 += rshr<4>(p2);
                            const double s3[3] = {qt[0] + readlane_d(p0, 7), qt[1] + readlane_d(p1, 7), qt[2] + readlane_d(p2, 7)};
                            sym3_mul(Bi, s3, zeta);
                        } else sym3_mul(Bi, qt, zeta);
                        zeta[0] = -zeta[0]; zeta[1] = -zeta[1]; zeta[2] = -zeta[2];
                        // a posteriori: the hard constraints must hold at the computed point (a last pivot of 1e-11 of its scale passes the test above,
                        // the multipliers are 1e17 and the point is noise: numerically singular -> the entering constraint is dependent)
                        if (has_p) {
                            double r = hy0 * zeta[0] + hy1 * zeta[1] + hy2 * zeta[2] - hrho;
                            if (ne > 0) {
#pragma unroll
                                for (int i = 0; i < R_NE; ++i) { if (i >= ne) continue; const double li = readlane_d(hlam, xb + i); if (ie >= 0 && ie < ne) r = fma(-omr[i], li, r); }
                            }
                            double wr = lane < nh ? fabs(r) : 0.0;   // (lanes 0 .. 7)
                            wr = max_raw(wr, rshr<1>(wr)); wr = max_raw(wr, rshr<2>(wr)); wr = max_raw(wr, rshr<4>(wr));
                            if (!(readlane_d(wr, 7) <= 1e-9)) sing = 1;
                        }
                    }
                    if (sing && phase != 2 && !blk) { rc = 2; why = 8; break; }
                    // dependent: the entering constraint in terms of the others, rr_c = (S_WW^-1 s)_c = its saved column over the rows' diagonals
                    if (sing) hlam = (lane < nh - 1) ? csave * idg : 0.0;
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R NEWVALS" ::: "memory");
#endif
                RPH(8);
                // ---- new values (regular) or rates per unit of the entering multiplier (dependent: the primal does not move)
                double a_n = a, mu_n = 0.0, lam_n = 0.0, lamp_n = 1.0, farkas = 0.0;
                // the hard rows' multipliers (or rates) back in their row lanes: list position of row `lane`
                const int hpos = (erow >= 0 && lane == erow) ? erl : (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));   // (set bits of hm below this lane: v_mbcnt, no lane mask kept in registers)
                double dlh = 0.0;
                if (sing) {   // rate of constraint c in lane c: the last one dl_last per unit of the entering multiplier, the others -rr dl_last
                    double dl_last = 1.0;
                    if (ent == RE_PIN0) dl_last = -readlane_d(risd, eidx);
                    if (ent == RE_PINL) dl_last = readlane_d(risd, eidx);
                    dlh = (lane < nh - 1) ? -hlam * dl_last : ((lane == nh - 1) ? dl_last : 0.0);
                    farkas = wave_sum0(dlh * hd);
                }
                const double hval = sing ? dlh : hlam;
                double wv[3];
#pragma unroll
                for (int x = 0; x < 3; ++x) wv[x] = fma(sg3[x], zeta[x], w03[x]);
                // the hard rows' values back in their row lanes (a handful of broadcasts: cheaper than a trip through the LDS crossbar), and
                // c = sum lam xi over the active rows: the hard rows' part summed over the lanes of the small system, the soft rows' part is
                // -(M_s w + m_s) (lam_j = -2/sd^2 (b + xi.w) - st/sd).  c is formed directly, not from zeta: on an axis whose components are all fixed
                // (g = 0) it moves no acceleration, but it is part of the gradient there, i.e. of the multipliers of those bounds.  (dependent: the
                // soft rows' multipliers do not move)
                double cv[3] = {0.0, 0.0, 0.0};
                if (nhr > 0) {
                    double lam_h = 0.0;
#pragma unroll
                    for (int c = 0; c < R_NH; ++c) {
                        if (c >= nhr0) break;
                        const double v = readlane_d(hval, c);
                        if (hpos == c) lam_h = v;
                    }
                    if (erow >= 0) { const double v = readlane_d(hval, erl); if (hpos == erl) lam_h = v; }
                    if (r_hard) lam_n = lam_h;
                    const bool mine = lane < nhr0 || lane == erl;
                    double p0 = mine ? hval * gx0 : 0.0, p1 = mine ? hval * gx1 : 0.0, p2 = mine ? hval * gx2 : 0.0;
                    p0 += rshr<1>(p0); p0 += rshr<2>(p0); p0 += rshr<4>(p0);
                    p1 += rshr<1>(p1); p1 += rshr<2>(p1); p1 += rshr<4>(p1);
                    p2 += rshr<1>(p2); p2 += rshr<2>(p2); p2 += rshr<4>(p2);
                    cv[0] = readlane_d(p0, 7); cv[1] = readlane_d(p1, 7); cv[2] = readlane_d(p2, 7);
                }
                if (!sing && softm_prev != 0ull) {
                    if (r_soft) lam_n = -(2.0 * ((rb + (xi0 * wv[0] + xi1 * wv[1] + xi2 * wv[2])) * risd) + st) * risd;
                    double mw[3];
                    sym3_mul(Ms, wv, mw);
                    cv[0] -= mw[0] + msv[0]; cv[1] -= mw[1] + msv[1]; cv[2] -= mw[2] + msv[2];
                }
                const double c_l = ax_l == 0 ? cv[0] : (ax_l == 1 ? cv[1] : cv[2]);
                double grad = -c_l * lkc;
                if (!sing) a_n = fma(Ykc, c_l, a0);
                if (ne > 0) {
                    if (ne == 1 && ent_extra && ent == RE_BOUND) {
                        const double le = readlane_d(hval, xb);
                        if (!sing) { a_n = fma(-le, Yp, a_n); lamp_n = le; }
                        if (lane == eidx) grad += le * (double)esg;
                    } else {
#pragma unroll
                        for (int i = 0; i < R_NE; ++i) {
                            if (i >= ne) continue;
                            const bool is_b = ent_extra && i == nw && ent == RE_BOUND;
                            const int code = i < nw ? wcode[i < R_NW ? i : 0] : (eidx | (esg > 0 ? 256 : 0));
                            const double nb = is_b ? ((lane == eidx) ? (double)esg : 0.0) : wall_normal(code, comp, ax_l, k_l, h2);
                            const double le = readlane_d(hval, xb + i);
                            if (!sing) { const double Yi = rax_solve(A, fixd ? 0.0 : nb, lKl, q2); a_n = fma(-le, Yi, a_n); }
                            grad = fma(le, nb, grad);
                            if (i < nw) { if (lane == 48 + i) mu_n = le; } else if (!sing) lamp_n = le;
                        }
                    }
                }
                if (!comp) a_n = 0.0;
                {
                    // (the product with H1 is a wave collective -- DPP row shifts and rotations read 0 from lanes that are switched off: never inside a lane-dependent branch)
                    const double ha_n = sing ? 0.0 : rax_hmul(a_n, dg, e_off, lKl, q2) + f_l;
                    if (comp && fx != 0) mu_n = -(double)fx * (ha_n + grad);
                }
                if (sing) farkas += wave_sum0((comp && fx != 0) ? mu_n * P.alim : 0.0);
#ifdef RSOLVE_MARK
                asm volatile("; @@R PHASE" ::: "memory");
#endif
                RPH(9);
                // =========================================================== what the phase does with it
#if defined(RSOLVE_TRACE)
                if (phase == 1 && P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 2) {
                    double *d = P.dbg + (size_t)(iters_total - 1) * 8;
                    d[0] = (double)(phase + 10 * ent + 1000 * eidx); d[1] = (double)(nh + 16 * ne + 256 * (sing + 1) + 4096 * nhr);
                    d[2] = w03[0]; d[3] = w03[1]; d[4] = w03[2]; d[5] = sg3[0] * sg3[0]; d[6] = sg3[1] * sg3[1]; d[7] = sg3[2] * sg3[2];
                }
#endif
                if (phase == 1) {   // crash: free the bounds whose multipliers came out negative, solve again; then the first scan
                    const unsigned long long neg = __ballot(comp && fx != 0 && mu_n < 0.0);
                    if (blk) {   // block move: the other multipliers must stay non-negative, or the move is taken back
                        const bool rneg = r_in && (lam_n < 0.0 || ((rfl & RB_PIN0) && -st - rsd * lam_n < 0.0) || ((rfl & RB_PINL) && fma(rsd, lam_n, 2.0 * slb + st) < 0.0));
                        const bool wneg = lane >= 48 && lane < 48 + nw && mu_n < 0.0;
                        if (sing || __ballot(rneg || wneg) != 0ull || ++inner > 12) {
                            RBLOCK_UNDO();
                            continue;
                        }
                    }
                    if (neg != 0ull) { if (comp && fx != 0 && mu_n < 0.0) fx = 0; fdirty = true; continue; }
                    a = a_n; mu = mu_n; lam = lam_n;
                    blk = false; noblock = false;
                    phase = 3;
                    continue;
                }
#ifdef RSOLVE_MARK
                asm volatile("; @@R RATIO" ::: "memory");
#endif
                RPH(9);
                // ---- ratio test over the multipliers of the working set (the entering constraint's own multiplier does not block)
                // inverse step lengths, identity 0: regular (new < 0): (cur - new) / cur >= 1 blocks at tau = cur / (cur - new); dependent (rate < 0): -rate / cur
                double ir = 0.0; int bt = -1;
#define RRT(cur_, new_, ty_) do { const double c__ = (cur_), n__ = (new_); if (n__ < 0.0) { \
                    const double num__ = sing ? -n__ : (c__ - n__); const double r__ = c__ > 1e-300 ? fast_div(num__, c__) : INFINITY; \
                    if (r__ > ir) { ir = r__; bt = (ty_); } } } while (0)
                if ((comp && fx != 0) || (lane >= 48 && lane < 48 + nw)) RRT(mu, mu_n, lane < 48 ? 0 : 4);
                if (r_in) {
                    const int own = (ent >= RE_ROW && ent <= RE_PINL && lane == eidx) ? ent : -1;
                    if (own != RE_ROW) RRT(lam, lam_n, 1);
                    // the pins' multipliers from lam: pi = -st - sd lam, rho = 2 slb + st + sd lam (rates: -sd dlam, +sd dlam)
                    if ((rfl & RB_PIN0) && own != RE_PIN0) RRT(-st - rsd * lam, sing ? -rsd * lam_n : -st - rsd * lam_n, 2);
                    if ((rfl & RB_PINL) && own != RE_PINL) RRT(fma(rsd, lam, 2.0 * slb + st), sing ? rsd * lam_n : fma(rsd, lam_n, 2.0 * slb + st), 3);
                }
#undef RRT
                const double imax = wave_max0(ir);
                const bool blocked = sing ? (imax > 0.0) : (imax > 1.0);
#if defined(RSOLVE_TRACE)
                if (P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 5) {   // development: one record per equality-constrained QP
                    double *d = P.dbg + (size_t)(iters_total - 1) * 8;
                    d[0] = (double)(phase + 10 * ent + 1000 * eidx); d[1] = (double)(nh + 16 * ne + 256 * (sing + 1) + 4096 * nhr);
                    d[2] = wv[0]; d[3] = wv[1]; d[4] = wv[2];
                    d[5] = imax; d[6] = sg3[0] * sg3[0]; d[7] = w03[0];
                }
#endif
#ifdef RSOLVE_MARK
                asm volatile("; @@R STEP" ::: "memory");
#endif
                RPH(10);
                if (sing && !blocked) {
                    if (!(farkas < 0.0)) { rc = 2; why = 9; break; }
                    // how far up the ladder does this Farkas combination reach (dmpc_solve.hip, round 5): C + 2^m U with U the part that carries slb
                    if (ladder && violation) {
                        const double u_l = (r_hard && (rfl & RB_PINL)) ? lam_n * (-rsd * slb) : 0.0;
                        const double Uc = wave_sum0(u_l), Cc = farkas - Uc;
                        if (Cc + Uc < 0.0 && !P.no_level_skip) {
                            double kk = 2.0;
                            while (lev_skip < 40 && Cc + kk * Uc < -1e-7 * (fabs(Cc) + kk * fabs(Uc))) { ++lev_skip; kk *= 2.0; }
                        }
                    }
                    rc = 1; break;
                }
                if (!blocked) {   // full step: the entering constraint joins the working set
                    a = a_n; mu = mu_n; lam = lam_n; noblock = false;
                    if (ent == RE_BOUND) { if (lane == eidx) { fx = esg; mu = lamp_n; } fdirty = true; }
                    else if (ent == RE_WALL) {
#pragma unroll
                        for (int i = 0; i < R_NW; ++i) if (i == nw) wcode[i] = eidx | (esg > 0 ? 256 : 0);
                        if (lane == 48 + nw) mu = lamp_n;
                        nw++;
                    }
                    phase = 3;
                    continue;
                }
                // the blocking multiplier
                const unsigned long long bm = __ballot(bt >= 0 && ir == imax);
                const int bl = __ffsll((long long)bm) - 1;
                const int bty = readlane_i(bt, bl);
                if (!sing && bty == 0 && !noblock && nblock < RBLOCK_MAX && leqp < RBLOCK_UNTIL && ent != RE_WALL) {
                    const bool ng = comp && fx != 0 && mu_n < 0.0;
                    if (__popcll(__ballot(ng)) >= 2) {   // block move (b): the entering constraint joins, every bound whose multiplier would turn negative is freed
                        fxhi_s = __ballot(comp && fx > 0); fxlo_s = __ballot(comp && fx < 0); rfl_s = rfl; sphase = 2;
                        if (ng) fx = 0;
                        if (ent == RE_BOUND && lane == eidx) fx = esg;
                        blk = true; ++nblock; fdirty = true; phase = 1; inner = 0;
                        continue;
                    }
                }
                // partial step to the blocking multiplier, which leaves the working set
                const double tau = imax < INFINITY ? fast_rcp(imax) : 0.0;
                if (!(tau > 0.0) && ++zero_steps > 6) { rc = 2; why = 10; break; }   // (degenerate: steps of length zero trading two dependent constraints for each other -- the general solver takes the agent)
                if (sing) { mu = fma(tau, mu_n, mu); lam = fma(tau, lam_n, lam); }
                else { a = fma(tau, a_n - a, a); mu = fma(tau, mu_n - mu, mu); lam = fma(tau, lam_n - lam, lam); }
#if defined(RSOLVE_TRACE)
                if (P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 5) { double *d = P.dbg + (size_t)(iters_total - 1) * 8; d[6] = (double)(bty * 100 + bl); d[7] = tau; }
#endif
                if (bty == 0) { if (lane == bl) { fx = 0; mu = 0.0; } fdirty = true; }
                else if (bty == 1) {
                    if (lane == bl) { rfl = RB_PIN0; lam = 0.0; }
                    if ((ent == RE_PIN0 || ent == RE_PINL) && eidx == bl) { phase = 3; continue; }   // the entering pin's row left: nothing to add
                }
                else if (bty == 2) { if (lane == bl) rfl &= ~RB_PIN0; }
                else if (bty == 3) { if (lane == bl) rfl &= ~RB_PINL; }
                else {   // wall bl - 48 leaves: the ones behind it move up
                    const int s = bl - 48;
                    const double m1 = readlane_d(mu, 49), m2 = readlane_d(mu, 50);
                    if (s == 0) { wcode[0] = wcode[1]; if (lane == 48) mu = m1; }
                    if (s <= 1) { wcode[1] = wcode[2]; if (lane == 49) mu = m2; }
                    nw--;
                }
            }
            {   // size of the working set in the general solver's terms: bounds + rows + instantiated pins + walls
                const int q = __popcll(__ballot(comp && fx != 0)) + __popcll(__ballot(rv && (rfl & RB_IN))) + __popcll(__ballot(rv && (rfl & RB_IN) && (rfl & (RB_PIN0 | RB_PINL)))) + nw;
                if (q > maxq) maxq = q;
                qfinal = q;
            }
            if (rc == 0) { solved = true; break; }
            if (rc == 2) { giveup = true; break; }
            // infeasible: the retry ladder (solveSoftDMPCbound.m:147-153): lb_eps *= 2, term *= 2
            if (ladder && violation) {
                double f = 2.0;
                while (tries < max_tries - 1) {
                    if (lev_skip > 0) { --lev_skip; f *= 2.0; ++tries; continue; }   // (infeasible by the proof the failed solve ended with)
                    if (!uni_b(ladder_level_infeasible(r_xi, r_b, r_sd, r_slb, r_kc, nr, B, P.h, P.alim, lev_f * f, RWALLS_STACKED(whi_l), RWALLS_STACKED(wlo_l), lane, RCERT_PLANES))) { cert_known = true; break; }
                    f *= 2.0; ++tries;
                }
                slb *= f; st *= f; lev_f *= f;
                continue;
            }
            if (ladder) tries = max_tries;
            break;
        }
        if (!solved && !giveup) status |= ST_INFEAS;
    }
#if defined(RSOLVE_TRACE)
    {
        const unsigned long long mhi = __ballot(comp && fx > 0), mlo = __ballot(comp && fx < 0), min_ = __ballot(rv && (rfl & RB_IN)), mp0 = __ballot(rv && (rfl & RB_IN) && (rfl & RB_PIN0)), mpl = __ballot(rv && (rfl & RB_PINL));
        if (P.dbg && gid == P.dbg_agent && lane == 0 && P.dbg_cap >= 2) {
            double *d = P.dbg + (size_t)(P.dbg_cap - 1) * 8;
            d[0] = (double)(mhi & 0xffffffffull); d[1] = (double)(mhi >> 32); d[2] = (double)(mlo & 0xffffffffull); d[3] = (double)(mlo >> 32);
            d[4] = (double)min_; d[5] = (double)mp0; d[6] = (double)mpl; d[7] = (double)(iters_total + 1000 * nw + 100000 * (giveup ? 1 : 0));
        }
    }
    if (P.dbg && gid == P.dbg_agent && lane == 0 && P.dbg_cap >= 4) { double *d = P.dbg + (size_t)(P.dbg_cap - 4) * 8; for (int u = 0; u < 12; ++u) d[u] = (double)rph[u]; }
    if (giveup && P.dbg && P.dbg_agent == -7 && lane == 0) { atomicAdd((int *)P.dbg + (why & 15), 1); P.dbg[16 + (why & 15)] = (double)gid + 1e-3 * (double)iters_total; }   // development: histogram of the reasons, an agent of each
#endif
    (void)why;
    const KargPtr Qp = kernarg_params();
    RCLAIM_NEXT();
    if (giveup) {   // the general kernel takes this agent (tier-2 launch over the flagged list)
        status = h0.w | ST_QOVER;
        if (lane == 0) {
            Qp->status[gid] = status;
            if (Qp->flag_list) Qp->flag_list[atomicAdd(Qp->flag_count, 1)] = gid;
        }
        return;
    }
    // ---------------------------------------------------------------- a9/a10: propagate, outputs (stacked order 3 k + axis through LDS)
    int nslack = 0;
    const bool no_set = solved && __ballot((comp && fx != 0) || (rv && (rfl & RB_IN))) == 0ull && nw == 0;
    if (solved) {
        status |= ST_SOLVED;
        const bool in_ = rv && (rfl & RB_IN);
        const double eps = !in_ ? 0.0 : ((rfl & RB_PINL) ? slb : ((rfl & RB_PIN0) ? 0.0 : -0.5 * fma(rsd, lam, st)));
        nslack = __popcll(__ballot(eps < -1e-12));
    }
    {
        const double s0 = row_prefix(comp ? a : 0.0), s1 = row_prefix(comp ? (double)k_l * a : 0.0);
        const double w = h2 * fma((double)k_l + 0.5, s0, -s1);
        LSYNC();
        if (comp) { B[3 * k_l + ax_l] = a; B[48 + 3 * k_l + ax_l] = w; }
        LSYNC();
    }
    double p_out = 0.0, v_out = 0.0, a_out = 0.0;
    const bool oc = lane < N3;
    const int ko = oc ? lane / 3 : 0, axo = oc ? lane - 3 * ko : 0;
    if (solved && oc) {
        typedef const double __attribute__((address_space(4))) *ConstD;
        const ConstD sp = (ConstD)(unsigned long long)(Qp->x_p + 3 * (size_t)gid), sv_ = (ConstD)(unsigned long long)(Qp->x_v + 3 * (size_t)gid);
        const double po0 = sp[0], po1 = sp[1], po2 = sp[2], vo0 = sv_[0], vo1 = sv_[1], vo2 = sv_[2];
        const double vo_o = axo == 0 ? vo0 : (axo == 1 ? vo1 : vo2);
        const double p0_o = init_pos(ko, Qp->h, vo_o, axo == 0 ? po0 : (axo == 1 ? po1 : po2));
        double w = B[48 + lane];
        if (no_set) {   // the unconstrained minimiser: its positions from the Gram table, bit for bit what the scan's unconstrained exit writes
            const double gx = goal_gap(axo == 0 ? Qp->pf[3 * (size_t)gid] : (axo == 1 ? Qp->pf[3 * (size_t)gid + 1] : Qp->pf[3 * (size_t)gid + 2]), axo == 0 ? po0 : (axo == 1 ? po1 : po2), vo_o, Qp->h);
            const double aoo = Qp->x_a[3 * (size_t)gid + axo];
            w = unc_entry(qw, sw, gx, aoo, Gt[(15 + ko) * 30 + 15 + (K - 1)], Gt[(15 + ko) * 30]);
        }
        p_out = w + p0_o;
        v_out = vel_out(B, ko, axo, Qp->h, vo_o);
        a_out = B[lane];
    }
    if (solved) {
        const bool ob_check = !cppv;
        if (h1.x & 4) status |= ST_COLL;
        if (ob_check) {
            const double tolb = 50e-3;
            bool bad = false;
            const double hi3 = lane == 0 ? Qp->pmax[0] : (lane == 1 ? Qp->pmax[1] : Qp->pmax[2]), lo3 = lane == 0 ? Qp->pmin[0] : (lane == 1 ? Qp->pmin[1] : Qp->pmin[2]);
            if (lane < 3) bad = !(p_out < hi3 + tolb) || !(p_out > lo3 - tolb);
            if (__any(bad)) status |= ST_OUTBOUND;
        }
    }
    if (oc) {
        Qp->p_out[(size_t)gid * N3 + lane] = p_out;
        Qp->v_out[(size_t)gid * N3 + lane] = v_out;
        Qp->a_out[(size_t)gid * N3 + lane] = a_out;
        if (Qp->lT_next) {
            const int Cq = Qp->C;
            const double *own = Qp->own_prev ? Qp->own_prev + (size_t)scene * N3 * Cq + cl : Qp->lT + ((size_t)(Qp->g_local * Qp->S + scene) * N3) * Cq + cl;
            Qp->lT_next[(size_t)scene * N3 * Cq + cl + (size_t)(unsigned)(lane * Cq)] = solved ? p_out : own[(size_t)(unsigned)(lane * Cq)];
        }
    }
    if (Qp->post_on) post_step_part(Qp, lane, gid, scene, solved, status, p_out, v_out, a_out);
    if (lane == 0) {
        Qp->status[gid] = status;
        if (Qp->cost_out) Qp->cost_out[gid] = cost;
        if (Qp->info) {
            int *inf = Qp->info + (size_t)gid * 8;
            inf[0] = viol_k; inf[1] = nrows_built; inf[2] = tries; inf[3] = (!solved && (status & ST_COLL)) ? 0 : ccase;
            inf[4] = iters_total; inf[5] = nslack; inf[6] = solved ? qfinal : 0; inf[7] = maxq;
        }
    }
#undef RCLAIM_NEXT
#undef RBLOCK_UNDO
#undef PMAXQ
#undef PMINQ
}
