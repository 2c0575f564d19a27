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
constexpr int R_NH = 5;   // hard constraints of the small system (hard rows + walls + the entering constraint)
constexpr int R_NE = 3;   // extras among them: two walls of the working set + an entering wall / bound

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

// one agent; the wave's 96 doubles of LDS (`smem`) serve the output stage only
__device__ __forceinline__ void rsolve_body(const StepParams &P, const int lane, const int vb, unsigned char *smem, const bool want_ticket, int &ticket, bool &claimed)
{
#define RCLAIM_NEXT() do { if (want_ticket && !claimed) { claimed = true; if (lane == 0) ticket = atomicAdd(kernarg_params()->counter, 1); } } while (0)
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
    double rb = r_b[ri], rsd = r_sd[ri], rst = r_st[ri], rslb = r_slb[ri];
    const int rkc = r_kc[ri];
    if (!rv) { xi0 = xi1 = xi2 = 0.0; rb = 0.0; rsd = 1.0; rst = 0.0; rslb = 0.0; }
    const int kc = (nr > 0) ? UNI(rkc) : 0;
    bool giveup = nr > 64 || __ballot(rv && rkc != kc) != 0ull;
    const double risd = fast_rcp(rsd);                                                  // 1 / sd
    const float rwt = 4.f * __builtin_amdgcn_rsqf((float)(xi0 * xi0 + xi1 * xi1 + xi2 * xi2));   // pivot weight of the row ("rows first", dmpc_solve.hip)
    const double rn2 = xi0 * xi0 + xi1 * xi1 + xi2 * xi2;

    // ---------------------------------------------------------------- cost case, component constants: lane = 16 axis + step
    const int ccase = UNI(cost_case(var, po[0] - pf[0], po[1] - pf[1], po[2] - pf[2], rows_exist));
    const double qw = ccase == 0 ? P.Qfar : (ccase == 1 ? P.Qnear : P.Q1);
    const double sw = ccase == 2 ? P.S1 : P.Sfree;
    const double q2 = 2.0 * qw, e_off = -2.0 * sw;
    const int ax_l = lane >> 4, k_l = lane & 15;
    const bool comp = ax_l < 3 && k_l < K;
    const double h2 = P.h * P.h;
    const double lKl = comp ? h2 * ((double)(K - 1 - k_l) + 0.5) : 0.0;                  // lK(k)
    const double lkc = (comp && k_l <= kc) ? h2 * ((double)(kc - k_l) + 0.5) : 0.0;      // l_kc(k)
    const double dg = comp ? (k_l < K - 1 ? 4.0 * sw + 2.0 : 2.0 * sw + 2.0) : 0.0;
    const double gax = comp ? goal_gap(sel3(pf, ax_l), sel3(po, ax_l), sel3(vo, ax_l), P.h) : 0.0;
    const double ao_l = comp ? sel3(ao, ax_l) : 0.0;
    const double f_l = comp ? (-q2 * lKl * gax - (k_l == 0 ? 2.0 * sw * ao_l : 0.0)) : 0.0;
    double whi_l = INFINITY, wlo_l = -INFINITY;
    if (comp) {
        const double sh = (double)(k_l + 1) * P.h * sel3(vo, ax_l);
        whi_l = sel3(P.pmax, ax_l) - sel3(po, ax_l) - sh;
        wlo_l = sel3(P.pmin, ax_l) - sel3(po, ax_l) - sh;
    }
    // unconstrained minimiser from the Gram tables, exactly as the scan's unconstrained exit and the general solver form it
    const double *Gt = P.tables + (size_t)ccase * TAB_CASE_DOUBLES;
    const int kt = comp ? k_l : 0;
    const double a_unc = comp ? unc_entry(qw, sw, gax, ao_l, Gt[kt * 30 + 15 + (K - 1)], Gt[kt * 30]) : 0.0;
    // scales of the dependence test: n'H^-1 n of the UNREDUCED Hessian (dmpc_solve.hip: delta <= 1e-13 s_pp)
    const double sc_bound_l = Gt[kt * 31];                 // H1^-1(k,k)
    const double sc_wall_l = Gt[(15 + kt) * 31];           // (L H1^-1 L')(k,k)
    const double sc_row = Gt[(15 + kc) * 31];

    const bool ladder = (var == VAR_BOUND || var == VAR_BOUND2 || cppv);
    const int max_tries = P.max_tries > 0 ? P.max_tries : (cppv ? 21 : 30);
    const double tol = 1e-10;
    int tries = h1.z, iters_total = 0, maxq = 0, qfinal = 0, cost = 0;
    bool solved = false;
    double a = 0.0, eps = 0.0;
    int fx = 0, rfl = 0, nw = 0;

    if (status & ST_INFEAS) tries = 1;
    if (tries > 0 && !(status & ST_INFEAS)) {
        if (tries >= max_tries) { status |= ST_INFEAS; tries = max_tries; }
        else { const double f = ldexp(1.0, tries); rslb *= f; rst *= f; }
    }
    if (!(status & (ST_COLL | ST_CAPACITY | ST_INFEAS)) && !giveup) {
        while (tries < max_tries) {
            tries++;
            int rc = 0;   // 0 solved, 1 infeasible, 2 give up
            int lev_skip = 0;
            // ---- state of the level
            fx = 0; rfl = rv ? RB_PIN0 : 0; nw = 0;
            a = a_unc; eps = 0.0;
            double mu = 0.0, lam = 0.0, pi_ = rv ? -rst : 0.0, rho = 0.0;
            int wcode[2] = {0, 0};          // walls of the working set: lane of the component | sign bit 8
            double lw[2] = {0.0, 0.0}, nbw[2] = {0.0, 0.0}, Yw[2] = {0.0, 0.0};
            RAx A;
            double a0 = 0.0, Ykc = 0.0;
            double g3[3] = {0, 0, 0}, w03[3] = {0, 0, 0}, sg3[3] = {0, 0, 0}, isg3[3] = {0, 0, 0};
            bool fdirty = true;
            int ent = RE_NONE, eidx = 0, esg = 0;   // entering constraint: type, lane of the component / row, sign
            double lam_p = 0.0;
            int phase = 0;                  // 0: crash start (fix the bounds violated at the unconstrained minimiser), 1: crash (free the negative multipliers), 2: iteration
            int inner = 0, iters = 0;
            {   // crash start
                const bool viol = comp && fabs(a_unc) - P.alim > tol;
                if (__ballot(viol) != 0ull) { fx = viol ? (a_unc > 0.0 ? 1 : -1) : 0; phase = 1; }
                else phase = 3;             // straight to the first violation scan: the unconstrained minimiser is the state
            }
            for (;;) {
                // =========================================================== violation scan (state: the minimiser of the working set)
                if (phase == 3) {
                    // positions w = Lambda a per axis by two prefix sums: w_k = h^2 ((k + 1/2) S0_k - S1_k)
                    const double s0 = row_prefix(comp ? a : 0.0), s1 = row_prefix(comp ? (double)k_l * a : 0.0);
                    const double w = h2 * fma((double)k_l + 0.5, s0, -s1);
                    const double wk0 = readlane_d(w, kc), wk1 = readlane_d(w, 16 + kc), wk2 = readlane_d(w, 32 + kc);
                    float bests = 0.f; int bestc = -1;
#define RCAND(v_, w_, code_) do { const double v__ = (v_); const float s__ = (float)v__ * (w_); if (v__ > tol && s__ > bests) { bests = s__; bestc = (code_); } } while (0)
                    if (comp) {
                        if (fx == 0) RCAND(fabs(a) - P.alim, 1.f, (RE_BOUND << 16) | (a > 0.0 ? 256 : 0) | lane);
                        const bool inw = (nw > 0 && (wcode[0] & 63) == lane) || (nw > 1 && (wcode[1] & 63) == lane);
                        const double c2 = w - whi_l, c3 = wlo_l - w;
                        if (!inw) RCAND(fmax(c2, c3), 1.f, (RE_WALL << 16) | (c2 > c3 ? 256 : 0) | lane);
                    }
                    if (rv) {
                        if (!(rfl & RB_IN)) RCAND(-(xi0 * wk0 + xi1 * wk1 + xi2 * wk2) - rb, rwt, (RE_ROW << 16) | lane);
                        else if (!(rfl & (RB_PIN0 | RB_PINL))) {
                            // (a lane is a component AND a row: the row's candidates compete with the component's through the same best-of)
                            RCAND(eps, 1.4142135f, (RE_PIN0 << 16) | lane);
                            RCAND(rslb - eps, 1.4142135f, (RE_PINL << 16) | lane);
                        }
                    }
#undef RCAND
                    const float smax = wave_max_f(bests);
                    const unsigned long long wm = __ballot(bestc >= 0 && bests == smax);
                    if (wm == 0ull) { RCLAIM_NEXT(); rc = 0; break; }   // optimal
                    if (++iters > P.iter_cap || iters > 600) { rc = 2; break; }
                    const int pcode = readlane_i(bestc, __ffsll((long long)wm) - 1);
                    ent = pcode >> 16; eidx = pcode & 63; esg = (pcode & 256) ? 1 : -1;
#if defined(DMPC_DEV_TRACE) || defined(RSOLVE_TRACE)
                    if (P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 3) {   // development: the scan's choice (shares the slot of the EQP that follows: written first, overwritten unless the EQP gives up)
                        double *d = P.dbg + (size_t)(P.dbg_cap - 2) * 8;
                        if (iters_total == 0) { d[0] = wk0; d[1] = wk1; d[2] = wk2; d[3] = (double)smax; d[4] = (double)pcode; d[5] = readlane_d(rb, eidx); d[6] = readlane_d(xi0, eidx); d[7] = readlane_d(xi1, eidx); }
                    }
#endif
                    if (ent == RE_WALL && nw >= 2) { rc = 2; break; }
                    // rows and pins join the working set at once with multiplier 0 (they stay "entering": their own multiplier does not block)
                    if (lane == eidx) {
                        if (ent == RE_ROW) rfl |= RB_IN;
                        if (ent == RE_PIN0) { rfl |= RB_PIN0; pi_ = 0.0; }
                        if (ent == RE_PINL) { rfl |= RB_PINL; rho = 0.0; }
                    }
                    lam_p = 0.0; inner = 0;
                    phase = 2;
                }
                if (phase == 2 && ++inner > 200) { rc = 2; break; }
                cost += 6;
                iters_total++;
                // =========================================================== the equality-constrained QP of the working set (+ entering constraint)
                const bool fixd = !comp || fx != 0;
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
                        g3[x] = readlane_d(gl, 16 * x); w03[x] = readlane_d(w0l, 16 * x);
                        const bool pos = g3[x] > 1e-300;
                        sg3[x] = pos ? g3[x] * fast_rsq(pos ? g3[x] : 1.0) : 0.0;
                        isg3[x] = pos ? fast_rsq(pos ? g3[x] : 1.0) : 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) if (i < nw) Yw[i] = rax_solve(A, fixd ? 0.0 : nbw[i], lKl, q2);
                }
                // ---- extras: the walls of the working set, then an entering wall / bound (a-space normal nbe, H~ nbe = Ye, both on one axis)
                int ne = nw;
                double nbe[R_NE] = {nbw[0], nbw[1], 0.0}, Ye[R_NE] = {Yw[0], Yw[1], 0.0};
                const bool ent_extra = (phase == 2) && (ent == RE_BOUND || ent == RE_WALL);
                if (nw == 1 && ent_extra) { nbe[1] = 0.0; Ye[1] = 0.0; }
                if (ent_extra) {
                    const int s = nw;   // its slot
                    double nbp;
                    if (ent == RE_BOUND) nbp = (lane == eidx) ? (double)esg : 0.0;
                    else {
                        const int ka = eidx & 15;
                        nbp = (comp && ax_l == (eidx >> 4) && k_l <= ka) ? (double)esg * h2 * ((double)(ka - k_l) + 0.5) : 0.0;
                    }
                    const double Yp = rax_solve(A, fixd ? 0.0 : nbp, lKl, q2);
#pragma unroll
                    for (int i = 0; i < R_NE; ++i) if (i == s) { nbe[i] = nbp; Ye[i] = Yp; }
                    ne = nw + 1;
                }
                // ---- soft rows: M_s = sum 2/sd^2 xi xi', q_s = M_s w0 + sum (2 b/sd^2 + st/sd) xi;  B = I + sqrt(G) M_s sqrt(G)
                const bool r_in = rv && (rfl & RB_IN), r_hard = r_in && (rfl & (RB_PIN0 | RB_PINL)), r_soft = r_in && !r_hard;
                Sym3 Bi; Bi.m00 = 1.0; Bi.m11 = 1.0; Bi.m22 = 1.0; Bi.m01 = 0.0; Bi.m02 = 0.0; Bi.m12 = 0.0;
                double qt[3] = {0.0, 0.0, 0.0};
                if (__ballot(r_soft) != 0ull) {
                    cost += 2;
                    const double al = r_soft ? 2.0 * risd * risd : 0.0;
                    const double be = r_soft ? (2.0 * rb * risd + rst) * risd : 0.0;
                    Sym3 Ms;
                    Ms.m00 = wave_sum0(al * xi0 * xi0); Ms.m01 = wave_sum0(al * xi0 * xi1); Ms.m02 = wave_sum0(al * xi0 * xi2);
                    Ms.m11 = wave_sum0(al * xi1 * xi1); Ms.m12 = wave_sum0(al * xi1 * xi2); Ms.m22 = wave_sum0(al * xi2 * xi2);
                    double qs[3];
                    sym3_mul(Ms, w03, qs);
                    qs[0] += wave_sum0(be * xi0); qs[1] += wave_sum0(be * xi1); qs[2] += wave_sum0(be * xi2);
                    Sym3 Bm;
                    Bm.m00 = fma(sg3[0] * sg3[0], Ms.m00, 1.0); Bm.m11 = fma(sg3[1] * sg3[1], Ms.m11, 1.0); Bm.m22 = fma(sg3[2] * sg3[2], Ms.m22, 1.0);
                    Bm.m01 = sg3[0] * sg3[1] * Ms.m01; Bm.m02 = sg3[0] * sg3[2] * Ms.m02; Bm.m12 = sg3[1] * sg3[2] * Ms.m12;
                    Bi = sym3_inv(Bm);
                    qt[0] = sg3[0] * qs[0]; qt[1] = sg3[1] * qs[1]; qt[2] = sg3[2] * qs[2];
                }
                // ---- the hard list, ONE CONSTRAINT PER LANE (lanes 0 .. nh-1): the hard rows of the working set in row order, the entering row
                // (while it is hard) last among them, then the extras.  Lane c holds yt_c = its normal in the scaled w-space, its right-hand sides
                // and row c of the small matrix S = Y' B^-1 Y + Om; the elimination broadcasts one pivot row at a time (Gauss-Jordan on a positive
                // definite matrix, the entering constraint LAST: its pivot is delta = n_p' P n_p, the dependence test of the dual method).
                int erow = (phase == 2 && ent >= RE_ROW && ent <= RE_PINL) ? eidx : -1;
                if (erow >= 0 && !((readlane_i(rfl, erow) & RB_IN) && (readlane_i(rfl, erow) & (RB_PIN0 | RB_PINL)))) erow = -1;   // soft by now (its pin gave way): no bordered constraint
                unsigned long long hm = __ballot(r_hard);
                if (erow >= 0) hm &= ~(1ull << erow);
                const int nhr0 = __popcll(hm);
                const int nhr = nhr0 + (erow >= 0 ? 1 : 0);
                const int nh = nhr + ne;
                if (nh > R_NH) { rc = 2; break; }
                const bool has_p = (phase == 2) && (ent_extra || erow >= 0);
                int src = 0;
                {
                    unsigned long long m = hm;
#pragma unroll
                    for (int c = 0; c < R_NH; ++c) {
                        if (c >= nhr0) continue;
                        const int j = __ffsll((long long)m) - 1; m &= m - 1ull;
                        if (lane == c) src = j;
                    }
                    if (erow >= 0 && lane == nhr0) src = erow;
                }
                double hy0, hy1, hy2, hrho = 0.0, hd = 0.0, hsc = 1.0;
                {
                    const double gx0 = __shfl(xi0, src), gx1 = __shfl(xi1, src), gx2 = __shfl(xi2, src), gb = __shfl(rb, src), gsd = __shfl(rsd, src), gslb = __shfl(rslb, src);
                    const int gfl = __shfl(rfl, src);
                    const bool mine = lane < nhr;
                    hy0 = mine ? -sg3[0] * gx0 : 0.0; hy1 = mine ? -sg3[1] * gx1 : 0.0; hy2 = mine ? -sg3[2] * gx2 : 0.0;
                    if (mine) {
                        hd = gb - ((gfl & RB_PINL) ? gsd * gslb : 0.0);
                        hrho = hd + (gx0 * w03[0] + gx1 * w03[1] + gx2 * w03[2]);
                        hsc = sc_row * (gx0 * gx0 + gx1 * gx1 + gx2 * gx2);
                    }
                }
                double om00 = 0.0, om01 = 0.0, om02 = 0.0, om11 = 0.0, om12 = 0.0, om22 = 0.0;   // Om = G0 - yt yt' over the extras
                if (ne > 0) {
                    cost += 2;
                    double yte[R_NE][3], g0[R_NE][R_NE];
#pragma unroll
                    for (int i = 0; i < R_NE; ++i) { yte[i][0] = yte[i][1] = yte[i][2] = 0.0; for (int j = 0; j < R_NE; ++j) g0[i][j] = 0.0; }
#pragma unroll
                    for (int i = 0; i < R_NE; ++i) {
                        if (i >= ne) continue;
                        const bool is_b = ent_extra && i == nw && ent == RE_BOUND;
                        const int wl = is_b ? eidx : ((i < nw ? wcode[i < 2 ? i : 0] : eidx) & 63);
                        const int wsg = is_b ? esg : (i < nw ? ((wcode[i < 2 ? i : 0] & 256) ? 1 : -1) : esg);
                        double t3[3], u0, dd, sc;
                        if (is_b) {   // unit normal: the sums are single entries
                            const double sgd = (double)esg;
                            const double t = sgd * readlane_d(Ykc, eidx);
                            const int x = eidx >> 4;
                            t3[0] = x == 0 ? t : 0.0; t3[1] = x == 1 ? t : 0.0; t3[2] = x == 2 ? t : 0.0;
                            u0 = sgd * readlane_d(a0, eidx); dd = P.alim; sc = Gt[(eidx & 15) * 31];
#pragma unroll
                            for (int j = 0; j <= i; ++j) g0[i][j] = g0[j][i] = sgd * readlane_d(Ye[j], eidx);
                        } else {
                            const double tl = row_allsum(lkc * Ye[i]), ul = row_allsum(nbe[i] * a0);
                            t3[0] = readlane_d(tl, 0); t3[1] = readlane_d(tl, 16); t3[2] = readlane_d(tl, 32);
                            u0 = readlane_d(ul, wl & 48);
                            dd = wsg > 0 ? readlane_d(whi_l, wl) : -readlane_d(wlo_l, wl);
                            sc = Gt[(15 + (wl & 15)) * 31];
#pragma unroll
                            for (int j = 0; j <= i; ++j) { const double gl2 = row_allsum(nbe[i] * Ye[j]); g0[i][j] = g0[j][i] = readlane_d(gl2, wl & 48); }
                        }
                        yte[i][0] = t3[0] * isg3[0]; yte[i][1] = t3[1] * isg3[1]; yte[i][2] = t3[2] * isg3[2];
                        if (lane == nhr + i) { hy0 = yte[i][0]; hy1 = yte[i][1]; hy2 = yte[i][2]; hd = dd; hrho = dd - u0; hsc = sc; }
                    }
                    om00 = g0[0][0] - (yte[0][0] * yte[0][0] + yte[0][1] * yte[0][1] + yte[0][2] * yte[0][2]);
                    om01 = g0[0][1] - (yte[0][0] * yte[1][0] + yte[0][1] * yte[1][1] + yte[0][2] * yte[1][2]);
                    om02 = g0[0][2] - (yte[0][0] * yte[2][0] + yte[0][1] * yte[2][1] + yte[0][2] * yte[2][2]);
                    om11 = g0[1][1] - (yte[1][0] * yte[1][0] + yte[1][1] * yte[1][1] + yte[1][2] * yte[1][2]);
                    om12 = g0[1][2] - (yte[1][0] * yte[2][0] + yte[1][1] * yte[2][1] + yte[1][2] * yte[2][2]);
                    om22 = g0[2][2] - (yte[2][0] * yte[2][0] + yte[2][1] * yte[2][1] + yte[2][2] * yte[2][2]);
                }
                // row c of S and the right-hand side in lane c
                double Sr[R_NH], rh, hlam = 0.0;
                double zeta[3] = {0.0, 0.0, 0.0};
                int sing = 0;
                {
                    const double hv[3] = {hy0, hy1, hy2};
                    double by[3], bq[3];
                    sym3_mul(Bi, hv, by);
                    sym3_mul(Bi, qt, bq);
                    rh = -hrho - (hy0 * bq[0] + hy1 * bq[1] + hy2 * bq[2]);
                    // my row of Om (extras i = lane - nhr)
                    const int ie = lane - nhr;
                    const double omr0 = ie == 0 ? om00 : (ie == 1 ? om01 : om02), omr1 = ie == 0 ? om01 : (ie == 1 ? om11 : om12), omr2 = ie == 0 ? om02 : (ie == 1 ? om12 : om22);
#pragma unroll
                    for (int e = 0; e < R_NH; ++e) {
                        Sr[e] = 0.0;
                        if (e >= nh) continue;
                        const double b0 = readlane_d(by[0], e), b1 = readlane_d(by[1], e), b2 = readlane_d(by[2], e);
                        double t = hy0 * b0 + hy1 * b1 + hy2 * b2;
                        if (e >= nhr && ie >= 0 && ie < ne) t += (e - nhr == 0) ? omr0 : ((e - nhr == 1) ? omr1 : omr2);
                        Sr[e] = (lane < nh) ? t : 0.0;
                    }
                    if (lane >= nh) rh = 0.0;
                    // Gauss-Jordan, pivots in list order; the entering constraint's pivot decides dependence
                    bool bad = false;
                    double csave = 0.0;   // column of the entering constraint in the eliminated rows, before its own pivot
#pragma unroll
                    for (int k = 0; k < R_NH; ++k) {
                        if (k >= nh) continue;
                        const double piv = readlane_d(Sr[k], k), psc = readlane_d(hsc, k);
                        if (k == nh - 1 && has_p) csave = Sr[k];
                        if (!(piv > 1e-13 * psc)) {
                            if (k == nh - 1 && has_p) sing = 1; else bad = true;
                            continue;
                        }
                        const double ipiv = fast_rcp(piv);
                        const double f = (lane == k) ? 0.0 : Sr[k] * ipiv;
                        const double prh = readlane_d(rh, k);
                        rh = fma(-f, prh, rh);
#pragma unroll
                        for (int e = 0; e < R_NH; ++e) {
                            if (e <= k || e >= nh) continue;
                            Sr[e] = fma(-f, readlane_d(Sr[e], k), Sr[e]);
                        }
                        if (lane != k) Sr[k] = 0.0;
                    }
                    if (uni_b(bad)) { rc = 2; break; }
                    sing = UNI(sing);
                    // my diagonal entry
                    const double dgn = lane == 0 ? Sr[0] : (lane == 1 ? Sr[1] : (lane == 2 ? Sr[2] : (lane == 3 ? Sr[3] : Sr[4])));
                    const double idg = (lane < nh && dgn > 0.0) ? fast_rcp(dgn) : 0.0;
                    if (!sing) {
                        hlam = rh * idg;
                        // zeta = -B^-1 (qt + sum_c yt_c lam_c): sums over the eight lanes that can hold a constraint
                        double p0 = hy0 * hlam, p1 = hy1 * hlam, p2 = hy2 * hlam;
                        p0 += rshr<1>(p0); p0 += rshr<2>(p0); p0 += rshr<4>(p0);
                        p1 += rshr<1>(p1); p1 += rshr<2>(p1); p1 += rshr<4>(p1);
                        p2 += rshr<1>(p2); p2 += rshr<2>(p2); p2 += rshr<4>(p2);
                        const double s3[3] = {qt[0] + readlane_d(p0, 7), qt[1] + readlane_d(p1, 7), qt[2] + readlane_d(p2, 7)};
                        sym3_mul(Bi, s3, zeta);
                        zeta[0] = -zeta[0]; zeta[1] = -zeta[1]; zeta[2] = -zeta[2];
                        // a posteriori: the hard constraints must hold at the computed point (a last pivot of 1e-11 of its scale passes the test above,
                        // the multipliers are 1e17 and the point is noise: numerically singular -> the entering constraint is dependent)
                        if (has_p && nh > 0) {
                            double r = hy0 * zeta[0] + hy1 * zeta[1] + hy2 * zeta[2] - hrho;
                            if (ne > 0) {
                                const double l0 = readlane_d(hlam, nhr), l1 = readlane_d(hlam, nhr + 1 < 64 ? nhr + 1 : 63), l2 = readlane_d(hlam, nhr + 2 < 64 ? nhr + 2 : 63);
                                if (ie >= 0 && ie < ne) r -= omr0 * l0 + (ne > 1 ? omr1 * l1 : 0.0) + (ne > 2 ? omr2 * l2 : 0.0);
                            }
                            const double worst = wave_max0(lane < nh ? fabs(r) : 0.0);
                            if (!(worst <= 1e-9)) sing = 1;
                        }
                    }
                    if (sing && phase != 2) { rc = 2; break; }
                    if (sing) {
                        // the entering constraint in terms of the others: rr_c = (S_WW^-1 s)_c = its column in the eliminated rows (saved before its own
                        // pivot) over their diagonal
                        hlam = (lane < nh - 1) ? csave * idg : 0.0;   // rr_c
                    }
                }
                // ---- new values (regular) or rates per unit of the entering multiplier (dependent: the primal does not move)
                double a_n = a, mu_n = 0.0, lam_n = 0.0, pi_n = 0.0, rho_n = 0.0, eps_n = eps, lamp_n = 1.0, farkas = 0.0;
                double lw_n[2] = {0.0, 0.0};
                // the hard rows' multipliers (or rates) back in their row lanes: list position of row `lane`
                const int hpos = (erow >= 0 && lane == erow) ? nhr0 : __popcll(hm & ((1ull << lane) - 1ull));
                if (!sing) {
                    const double lam_h = __shfl(hlam, r_hard ? hpos : 0);
                    double wv[3];
#pragma unroll
                    for (int x = 0; x < 3; ++x) wv[x] = fma(sg3[x], zeta[x], w03[x]);
                    if (rv) {
                        pi_n = -rst;
                        if (r_soft) {
                            eps_n = (rb + (xi0 * wv[0] + xi1 * wv[1] + xi2 * wv[2])) * risd;
                            lam_n = -(2.0 * eps_n + rst) * risd;
                            pi_n = 0.0;
                        } else if (r_hard) {
                            const bool low = (rfl & RB_PINL) != 0;
                            eps_n = low ? rslb : 0.0;
                            lam_n = lam_h;
                            if (low) { pi_n = 0.0; rho_n = fma(rsd, lam_n, 2.0 * rslb + rst); }
                            else pi_n = -rst - rsd * lam_n;
                        } else eps_n = 0.0;
                    }
                    // c = sum lam xi over the active rows -- summed directly: on an axis whose components are all fixed (g = 0) c moves no
                    // acceleration, but it is part of the gradient there, i.e. of the multipliers of those bounds
                    double cv[3] = {0.0, 0.0, 0.0};
                    if (__ballot(r_in) != 0ull) {
                        const double lr = r_in ? lam_n : 0.0;
                        cv[0] = wave_sum0(lr * xi0); cv[1] = wave_sum0(lr * xi1); cv[2] = wave_sum0(lr * xi2);
                    }
                    const double c_l = ax_l == 0 ? cv[0] : (ax_l == 1 ? cv[1] : cv[2]);
                    a_n = fma(Ykc, c_l, a0);
                    double grad = -c_l * lkc;
#pragma unroll
                    for (int i = 0; i < R_NE; ++i) {
                        if (i >= ne) continue;
                        const double le = readlane_d(hlam, nhr + i);
                        a_n = fma(-le, Ye[i], a_n);
                        grad = fma(le, nbe[i], grad);
                        if (i < nw) lw_n[i < 2 ? i : 0] = le; else lamp_n = le;
                    }
                    if (!comp) a_n = 0.0;
                    // (the product with H1 is a wave collective -- DPP row shifts and rotations read 0 from lanes that are switched off: never inside a lane-dependent branch)
                    const double ha_n = rax_hmul(a_n, dg, e_off, lKl, q2);
                    mu_n = (comp && fx != 0) ? -(double)fx * (ha_n + f_l + grad) : 0.0;
                } else {
                    // dl of the last hard entry per unit of the entering multiplier, the others: -rr dl_last
                    double dl_last = 1.0;
                    if (ent == RE_PIN0) dl_last = -readlane_d(risd, eidx);
                    if (ent == RE_PINL) dl_last = readlane_d(risd, eidx);
                    const double dlh = (lane < nh - 1) ? -hlam * dl_last : ((lane == nh - 1) ? dl_last : 0.0);   // lane c: rate of constraint c
                    double fk = wave_sum0(dlh * hd);
                    const double dl_row = __shfl(dlh, r_hard ? hpos : 0);
                    if (rv && r_hard) {
                        lam_n = dl_row;
                        if (rfl & RB_PINL) rho_n = rsd * lam_n; else pi_n = -rsd * lam_n;
                    }
                    double dc[3] = {0.0, 0.0, 0.0};
                    {
                        const double lr = (rv && r_hard) ? lam_n : 0.0;
                        dc[0] = wave_sum0(lr * xi0); dc[1] = wave_sum0(lr * xi1); dc[2] = wave_sum0(lr * xi2);
                    }
                    const double c_l = ax_l == 0 ? dc[0] : (ax_l == 1 ? dc[1] : dc[2]);
                    double grad = -c_l * lkc;
#pragma unroll
                    for (int i = 0; i < R_NE; ++i) {
                        if (i >= ne) continue;
                        const double le = readlane_d(dlh, nhr + i);
                        grad = fma(le, nbe[i], grad);
                        if (i < nw) lw_n[i < 2 ? i : 0] = le;
                    }
                    mu_n = (comp && fx != 0) ? -(double)fx * grad : 0.0;
                    fk += wave_sum0(mu_n * P.alim);
                    farkas = fk;
                    lamp_n = 1.0;
                }
                // =========================================================== what the phase does with it
#if defined(DMPC_DEV_TRACE) || defined(RSOLVE_TRACE)
                if (phase == 1 && P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 2) {
                    double *d = P.dbg + (size_t)(iters_total - 1) * 8;
                    d[0] = (double)(phase + 10 * ent + 1000 * eidx); d[1] = (double)(nh + 16 * ne + 256 * (sing + 1) + 4096 * nhr);
                    d[2] = w03[0]; d[3] = w03[1]; d[4] = w03[2]; d[5] = g3[0]; d[6] = g3[1]; d[7] = g3[2];
                }
#endif
                if (phase == 1) {   // crash: free the bounds whose multipliers came out negative, solve again; then the first scan
                    const unsigned long long neg = __ballot(comp && fx != 0 && mu_n < 0.0);
                    if (neg != 0ull) { if (comp && fx != 0 && mu_n < 0.0) fx = 0; fdirty = true; continue; }
                    a = a_n; mu = mu_n; eps = eps_n; lam = lam_n; pi_ = pi_n; rho = rho_n;
                    phase = 3;
                    continue;
                }
                // ---- ratio test over the multipliers of the working set (the entering constraint's own multiplier does not block)
                // inverse step lengths, identity 0: regular (new < 0): (cur - new) / cur >= 1 blocks at tau = cur / (cur - new); dependent (rate < 0): -rate / cur
                double ir = 0.0; int bt = -1;
#define RRT(cur_, new_, ty_) do { const double c__ = (cur_), n__ = (new_); if (n__ < 0.0) { \
                    const double num__ = sing ? -n__ : (c__ - n__); const double r__ = c__ > 1e-300 ? fast_div(num__, c__) : INFINITY; \
                    if (r__ > ir) { ir = r__; bt = (ty_); } } } while (0)
                if (comp && fx != 0) RRT(mu, mu_n, 0);
                if (r_in) {
                    const int own = (phase == 2 && ent >= RE_ROW && ent <= RE_PINL && lane == eidx) ? ent : -1;
                    if (own != RE_ROW) RRT(lam, lam_n, 1);
                    if ((rfl & RB_PIN0) && own != RE_PIN0) RRT(pi_, pi_n, 2);
                    if ((rfl & RB_PINL) && own != RE_PINL) RRT(rho, rho_n, 3);
                }
                // (walls: uniform values, tested by lanes 48 and 49 -- never component lanes with a multiplier of their own)
                if (lane == 48 && nw > 0) RRT(lw[0], lw_n[0], 4);
                if (lane == 49 && nw > 1) RRT(lw[1], lw_n[1], 5);
#undef RRT
                const double imax = wave_max0(ir);
                const bool blocked = sing ? (imax > 0.0) : (imax > 1.0);
#if defined(DMPC_DEV_TRACE) || defined(RSOLVE_TRACE)
                if (P.dbg && gid == P.dbg_agent && lane == 0 && iters_total <= P.dbg_cap - 2) {   // development: one record per equality-constrained QP
                    double *d = P.dbg + (size_t)(iters_total - 1) * 8;
                    d[0] = (double)(phase + 10 * ent + 1000 * eidx); d[1] = (double)(nh + 16 * ne + 256 * (sing + 1) + 4096 * nhr);
                    d[2] = fma(sg3[0], zeta[0], w03[0]); d[3] = fma(sg3[1], zeta[1], w03[1]); d[4] = fma(sg3[2], zeta[2], w03[2]);
                    d[5] = imax; d[6] = g3[0]; d[7] = w03[0];
                }
#endif
                if (sing && !blocked) {
                    if (!(farkas < 0.0)) { rc = 2; break; }
                    // how far up the ladder does this Farkas combination reach (dmpc_solve.hip, round 5): C + 2^m U with U the part that carries slb
                    if (ladder && violation) {
                        const double u_l = (rv && r_hard && (rfl & RB_PINL)) ? lam_n * (-rsd * rslb) : 0.0;
                        const double Uc = wave_sum0(u_l), Cc = farkas - Uc;
                        if (Cc + Uc < 0.0) {
                            double kk = 2.0;
                            while (lev_skip < 40 && Cc + kk * Uc < -1e-7 * (fabs(Cc) + kk * fabs(Uc))) { ++lev_skip; kk *= 2.0; }
                        }
                    }
                    rc = 1; break;
                }
                if (!blocked) {   // full step: the entering constraint joins the working set
                    a = a_n; mu = mu_n; eps = eps_n; lam = lam_n; pi_ = pi_n; rho = rho_n; lw[0] = lw_n[0]; lw[1] = lw_n[1];
                    if (ent == RE_BOUND) { if (lane == eidx) { fx = esg; mu = lamp_n; } fdirty = true; }
                    else if (ent == RE_WALL) {
                        const int s = nw;
#pragma unroll
                        for (int i = 0; i < 2; ++i) if (i == s) { wcode[i] = eidx | (esg > 0 ? 256 : 0); lw[i] = lamp_n; nbw[i] = nbe[i]; Yw[i] = Ye[i]; }
                        nw++;
                    }
                    phase = 3;
                    continue;
                }
                // partial step to the blocking multiplier, which leaves the working set
                const double tau = imax < INFINITY ? fast_rcp(imax) : 0.0;
                if (sing) {
                    mu = fma(tau, mu_n, mu); lam = fma(tau, lam_n, lam); pi_ = fma(tau, pi_n, pi_); rho = fma(tau, rho_n, rho);
                    lw[0] = fma(tau, lw_n[0], lw[0]); lw[1] = fma(tau, lw_n[1], lw[1]);
                    lam_p += tau;
                } else {
                    a = fma(tau, a_n - a, a); eps = fma(tau, eps_n - eps, eps);
                    mu = fma(tau, mu_n - mu, mu); lam = fma(tau, lam_n - lam, lam); pi_ = fma(tau, pi_n - pi_, pi_); rho = fma(tau, rho_n - rho, rho);
                    lw[0] = fma(tau, lw_n[0] - lw[0], lw[0]); lw[1] = fma(tau, lw_n[1] - lw[1], lw[1]);
                    lam_p = fma(tau, lamp_n - lam_p, lam_p);
                }
                const unsigned long long bm = __ballot(bt >= 0 && ir == imax);
                const int bl = __ffsll((long long)bm) - 1;
                const int bty = readlane_i(bt, bl);
                if (bty == 0) { if (lane == bl) { fx = 0; mu = 0.0; } fdirty = true; }
                else if (bty == 1) {
                    if (lane == bl) { rfl = RB_PIN0; lam = 0.0; pi_ = -rst; rho = 0.0; eps = 0.0; }
                    if ((ent == RE_PIN0 || ent == RE_PINL) && eidx == bl) { phase = 3; continue; }   // the entering pin's row left: nothing to add
                }
                else if (bty == 2) { if (lane == bl) { rfl &= ~RB_PIN0; pi_ = 0.0; } }
                else if (bty == 3) { if (lane == bl) { rfl &= ~RB_PINL; rho = 0.0; } }
                else {
                    const int s = bty - 4;
                    if (s == 0) { wcode[0] = wcode[1]; lw[0] = lw[1]; nbw[0] = nbw[1]; Yw[0] = Yw[1]; }
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
                while (tries < max_tries - 1 && lev_skip > 0) { --lev_skip; f *= 2.0; ++tries; }
                rslb *= f; rst *= f;
                continue;
            }
            if (ladder) tries = max_tries;
            break;
        }
        if (!solved && !giveup) status |= ST_INFEAS;
    }
#if defined(DMPC_DEV_TRACE) || defined(RSOLVE_TRACE)
    {
        const unsigned long long mhi = __ballot(comp && fx > 0), mlo = __ballot(comp && fx < 0), min_ = __ballot(rv && (rfl & RB_IN)), mp0 = __ballot(rv && (rfl & RB_IN) && (rfl & RB_PIN0)), mpl = __ballot(rv && (rfl & RB_PINL));
        if (P.dbg && gid == P.dbg_agent && lane == 0 && P.dbg_cap >= 2) {
            double *d = P.dbg + (size_t)(P.dbg_cap - 1) * 8;
            d[0] = (double)(mhi & 0xffffffffull); d[1] = (double)(mhi >> 32); d[2] = (double)(mlo & 0xffffffffull); d[3] = (double)(mlo >> 32);
            d[4] = (double)min_; d[5] = (double)mp0; d[6] = (double)mpl; d[7] = (double)(iters_total + 1000 * nw + 100000 * (giveup ? 1 : 0));
        }
    }
#endif
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
        nslack = (int)wave_sum0((rv && eps < -1e-12) ? 1.0 : 0.0);
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
}
