// dmpc_solve.hip -- the QP phase of one MPC step (a7-a10): included by dmpc_kernels.hip inside namespace dmpc.
//
//   solveSoftDMPCbound.m:43-160 and the seven sibling solvers: cost case, (H, f), the QP by a dual active-set
//   (Goldfarb-Idnani) method in Schur-complement form, the retry ladder, propStatedmpc.m:3-4, is_inbounds.m:2-5.
//
// One 64-lane wave owns one agent.  Round-2 form of the iteration (the kernel is instruction-issue bound, so the
// instruction count per active-set iteration is what this layout is built around):
//   * the working-set capacity QCAP is a template parameter: every LDS object of the solver sits at a compile-time
//     offset from the wave's base (immediate offsets in the ds_ instructions, no address arithmetic, few live SGPRs);
//   * ONE symmetric 30x30 "Gram" table G per cost case over the index (space, step): space A = acceleration
//     components, space W = position components:  G[A i][A j] = H1^-1(i,j), G[A i][W j] = (H1^-1 L')(i,j),
//     G[W i][W j] = (L H1^-1 L')(i,j).  n_i' H^-1 n_j of any two constraints is G[gi_i][gi_j] x a 3-dot, with no
//     case distinction on the constraint types; z = H^-1 nu and Lambda z are rows k and 15+k of G against nu;
//   * the inverse factor T keeps the padded column-major layout with the odd column stride; every LDS value of a wave
//     is finite (the region is zeroed once), vectors are exactly zero beyond q, so T x and T'x run in unmasked groups
//     of 8 with ONE lane mask per group;
//   * box and workspace bounds of one component are mutually exclusive pairs (a <= alim and -a <= alim are never
//     active together): one candidate and one slot lookup per pair;
//   * collision slots enter the residual through a per-lane mask of the slots that constrain the lane's own horizon
//     step (a loop of max-slots-per-step rounds instead of one round per collision slot of the working set);
//   * the pivot score is compared in fp32 (one DPP max per step instead of a two-register fp64 butterfly); the
//     violation, the step lengths and every quantity that reaches the result stay fp64.
// All arithmetic that reaches an output is fp64; no atomics; fixed-order reductions => bit-reproducible.

// TS < QCAP ("split T", persistent slack-free kernels, round 4): the wave's block holds columns 0 .. TS-1 of T only.  The fp64 inverse
// factor was 10.9 of the 15.1 KB of LDS per wave and capped the launch at nine waves per CU, while 97 % of the agents of the headline
// launch never hold more than 16 constraints.  An agent that outgrows TS columns takes an EXTENSION (columns TS .. QCAP-1 in the same
// padded layout) from a pool the workgroup shares, at the wave-uniform offset `xo` doubles from the wave's base (0: none held): the same
// values at other addresses -- results are bit for bit those of the unsplit layout.
// TF = float (DMPC_PREC_F32FACTOR, round 4: the "QP below fp64" of BASELINE configs[4]): the inverse factor is STORED in fp32 -- half the
// LDS of the solver's largest object; products, rotations and every other quantity stay fp64, and the refinement pass (lambda += T T' rho
// against fp64 residuals) runs until the active-set residual is back at 1e-13 or stops contracting.
template <bool SOFT, int QCAP, bool PERSIST, int TS = QCAP, typename TF = double>
struct SolveLds {   // offsets in doubles from the wave's LDS base
    static constexpr int T = 0;
    static constexpr int A = (t_doubles(TS) * (int)sizeof(TF) + 7) / 8;
    static constexpr int W = A + 48;
    static constexpr int RR = W + 48;
    static constexpr int XS = RR + 64;
    // slack-free variants: Y lives where x (the input of T'x) was, nu where r was -- both are dead by then; the slack
    // bookkeeping of the soft variants still reads r after nu is written, so they keep separate vectors
    // (round 4: Y over x for the slack variants too -- what their bookkeeping still reads after nu is written is r, not x; the 48 doubles pay
    // for the rows' slot maps, so that eight slack waves still fit a CU next to the tables)
    static constexpr int Y = XS;
    static constexpr int NU = SOFT ? XS + 64 : RR;
    static constexpr int SVEC = SOFT ? NU + 48 : XS + 64;
    static constexpr int SD = SVEC + 3 * QCAP;
    static constexpr int SLAM = SD + QCAP;
    static constexpr int SSS = SLAM + QCAP;
    static constexpr int META = SSS + (SOFT ? QCAP : 0);   // QCAP ints
    static constexpr int WU = META + QCAP / 2;              // slack-free variants: w_unc (48 doubles; two registers less across the solver loop)
    static constexpr int TAB = WU + (SOFT ? 0 : 48);        // one-agent-per-workgroup form: own copy of G (900) + Lt (225)
    static constexpr int VAR = TAB + (PERSIST ? 0 : TAB_CASE_DOUBLES + TAB_L_DOUBLES + 1);   // r_eps (soft: nrmax doubles), r_fl (nrmax bytes)
};

// wave-wide maximum of non-NaN floats: v_max_f32 with the DPP operand folded in (lanes without a source keep their
// value), two wait states between a VALU write and the DPP read of the same register
__device__ __forceinline__ float wave_max_f(float v)
{
    // (the FIRST wait is five states, not two: the assembler block may directly follow an s_or_b64 exec that closes a lane-dependent branch -- the
    // candidate blocks of the violation scans -- and a DPP operation needs five wait states behind a write of EXEC; the hazard recogniser does not
    // look into inline assembly.  dmpc_rsolve.hip, round 6: with the stale mask the maximum missed lanes, no lane equalled it, and the scan
    // reported "nothing violated" -- in one build of two)
    asm volatile("s_nop 4\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Wave-wide sum / maximum of NON-NEGATIVE-identity reductions with bound_ctrl DPP moves: a lane without a source reads 0,
// which is the identity of the sum (and of the maximum of non-negative values), so no identity register has to be
// initialised per step (2 of the 5 instructions of a step in the generic form of dmpc_kernels.hip).  All rows take part in
// the two row_bcast steps; lane 63 ends with (r3 + r2) + (r0 + r1).  Fixed order => bit-reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp0_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum0(double v)
{
    v += dpp0_d<0x111>(v); v += dpp0_d<0x112>(v); v += dpp0_d<0x114>(v); v += dpp0_d<0x118>(v);
    v += dpp0_d<0x142>(v); v += dpp0_d<0x143>(v);
    return readlane_d(v, 63);
}
// Reciprocal, quotient and reciprocal square root of NORMAL positive doubles in a handful of instructions: the hardware estimate
// (v_rcp_f64 / v_rsq_f64, ~26 bits) and two Newton steps (to the last bit or two), the quotient with one correction step on top.
// The IEEE forms the compiler emits for `/`, `sqrt`, `rsqrt` carry 30-35 instructions each (scaling for denormals, the special
// values, exact rounding): an eighth of an active-set iteration went into three quotients and one root.  The callers keep 0 and inf out
// (explicit selects); nothing here reaches a result directly -- step lengths, the ratio test, the new factor column; the iterate the
// solver returns is re-derived from the multipliers and refined.
__device__ __forceinline__ double fast_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    return fma(x, e, x);
}
__device__ __forceinline__ double fast_div(double a, double b)
{
    const double x = fast_rcp(b), q = a * x;
    return fma(fma(-b, q, a), x, q);
}
__device__ __forceinline__ double fast_rsq(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-d * y, y, 1.0);
    return fma(0.5 * y, e, y);
}
// inclusive prefix sum over the lanes (lane j: v_0 + ... + v_j), the gfx9 DPP scan: row_shr 1/2/4/8 inside the 16-lane rows, then
// the two row broadcasts (rows 1 and 3 take row 0's / row 2's total, rows 2 and 3 the total of the lower half).  Fixed order.
__device__ __forceinline__ double wave_scan_sum(double v)
{
    v += dpp_d<0x111, 0xf>(0.0, v); v += dpp_d<0x112, 0xf>(0.0, v); v += dpp_d<0x114, 0xf>(0.0, v); v += dpp_d<0x118, 0xf>(0.0, v);
    v += dpp_d<0x142, 0xa>(0.0, v); v += dpp_d<0x143, 0xc>(0.0, v);
    return v;
}
// v_max_f64 without the canonicalising self-maximum the compiler puts in front of fmax() (inputs are never NaN here)
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_max0(double v)   // v >= 0 on every lane
{
    v = max_raw(v, dpp0_d<0x111>(v)); v = max_raw(v, dpp0_d<0x112>(v)); v = max_raw(v, dpp0_d<0x114>(v));
    v = max_raw(v, dpp0_d<0x118>(v)); v = max_raw(v, dpp0_d<0x142>(v)); v = max_raw(v, dpp0_d<0x143>(v));
    return readlane_d(v, 63);
}

// wave-uniform by construction, not by what the compiler can see (a value read from LDS or global memory at a uniform address, the result
// of an out-of-line function): say so -- everything derived from it is then scalar, branches on it are scalar branches
#define UNI(x_) __builtin_amdgcn_readfirstlane(x_)
__device__ __forceinline__ bool uni_b(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// uniform description of one constraint
struct Cd {
    int ty, idx, gi, si;     // type, index (component or row), Gram index (space, step), slack row (-1: none)
    double v0, v1, v2, ss, d;
};

typedef __attribute__((address_space(3))) double LdsD;
// y = T' x  (lane j gets y_j; 0 for j >= q).  x is the LDS vector at offset XOFF (zero beyond q).  Groups of 8 with one
// lane mask per group; two FMA chains per group (half the dependent latency).
// (round 5: the group loop fully unrolled with a scalar exit per group -- every LDS offset of a group is then an immediate of its
// ds_read; the rolled loop carried 6 scalar induction variables and 7 vector address additions per group of T x, a third of its
// instructions.  Same sums in the same order: bit-identical.)
template <int QCAP, int TOFF, int XOFF, int TS = QCAP, typename TF = double, bool PAIR = false>
__device__ __forceinline__ double t_tmul2(const double *B, int lane, int q, const int xo = 0)
{
    double acc = 0.0;
    const int qlim = (TS < QCAP && xo == 0) ? TS : QCAP;   // (no extension held: q <= TS, the lanes beyond read column TS-1, masked)
    const int jc = lane < qlim ? lane : qlim - 1;
    const TF *col = (const TF *)(B + TOFF) + tcol(jc) + ((TS < QCAP && jc >= TS) ? xo : 0);
    // (x through ONE vector register with immediate offsets, the group masks compared on the spot: as wave-uniform / loop-invariant values
    // the compiler computed every address and every mask once per solve, kept them in scalar registers, spilled those to vector lanes and
    // read them back in front of each use -- three instructions where one does)
    const LdsD *xs = (const LdsD *)(B + XOFF);
    asm volatile("" : "+v"(xs));
    asm volatile("" : "+v"(lane));
    // (two groups per round trip: the loads of both are issued before either is summed.  The second group may lie beyond q: x is exactly zero
    // there and every value of the wave's LDS is finite, so it adds +0 -- the same bits as leaving it out)
    // (PAIR: the slack kernels, which have the registers -- 2 waves per SIMD; the slack-free kernels sit at their 168 and would spill)
    constexpr int GS = PAIR ? 16 : 8;
#pragma unroll
    for (int i0 = 0; i0 < QCAP; i0 += GS) {
        if (i0 >= q) break;
        constexpr int NG = GS / 8;
        double t[GS], x[GS];
#pragma unroll
        for (int u = 0; u < GS; ++u) { const bool in = i0 + u < QCAP; t[u] = in ? (double)col[i0 + u] : 0.0; x[u] = in ? xs[i0 + u] : 0.0; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (i0 + 8 * g >= QCAP) break;
            double s0 = t[8 * g] * x[8 * g], s1 = t[8 * g + 1] * x[8 * g + 1];
#pragma unroll
            for (int u = 2; u < 8; u += 2) { s0 = fma(t[8 * g + u], x[8 * g + u], s0); s1 = fma(t[8 * g + u + 1], x[8 * g + u + 1], s1); }
            acc += (i0 + 8 * g <= lane) ? (s0 + s1) : 0.0;
        }
    }
    return (lane < q) ? acc : 0.0;
}
// y = T x  (lane i gets y_i; 0 for i >= q).  The 8 columns of group g hold rows 0 .. 8(g+1)-1 (zeros below the diagonal),
// so a lane either owns the whole group or skips it; columns >= q only meet x_j = 0.
template <int QCAP, int TOFF, int XOFF, int TS = QCAP, typename TF = double, bool PAIR = false>
__device__ __forceinline__ double t_mul2(const double *B, int lane, int q, const int xo = 0)
{
    double acc = 0.0;
    const TF *row = (const TF *)(B + TOFF) + lane;
    const TF *rowx = row + xo;   // the column groups of the extension (split T; the same sums in the same order)
    const LdsD *xs = (const LdsD *)(B + XOFF);
    asm volatile("" : "+v"(xs));
    asm volatile("" : "+v"(lane));
    constexpr int GS = PAIR ? 16 : 8;
#pragma unroll
    for (int j0 = 0; j0 < QCAP; j0 += GS) {
        if (j0 >= q) break;
        constexpr int NG = GS / 8;
        double t[GS], x[GS];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int jg = j0 + 8 * g;
            if (jg >= QCAP) break;
            // (split T without an extension held: the group at TS reads the vectors behind the wave's columns -- finite -- against x = 0)
            const TF *rw = (TS < QCAP && jg >= TS) ? rowx : row;
            const int c0 = tcol(jg), len = jg + 9;
#pragma unroll
            for (int u = 0; u < 8; ++u) { t[8 * g + u] = (double)rw[c0 + u * len]; x[8 * g + u] = xs[jg + u]; }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int jg = j0 + 8 * g;
            if (jg >= QCAP) break;
            double s0 = t[8 * g] * x[8 * g], s1 = t[8 * g + 1] * x[8 * g + 1];
#pragma unroll
            for (int u = 2; u < 8; u += 2) { s0 = fma(t[8 * g + u], x[8 * g + u], s0); s1 = fma(t[8 * g + u + 1], x[8 * g + u + 1], s1); }
            acc += (lane < jg + 8) ? (s0 + s1) : 0.0;
        }
    }
    return (lane < q) ? acc : 0.0;
}

// delete slot l: Givens rotations on adjacent columns zero row l of T left-to-right; row l and the last column drop out.
// Fused with the row deletion (each lane carries its row of the "current right column" in a register).
template <bool SOFT, int QCAP, bool PERSIST, int TS = QCAP, typename TF = double>
__device__ __forceinline__ void remove_slot2(double *B, int lane, int &q, int l, unsigned &cslot, unsigned long long &cm, const int xo = 0,
                                             unsigned char *rmap = nullptr /* slack kernels: the rows' slot maps [3][nrm] (collision / eps <= 0 / eps >= slb) */, const int nrm = 0)
{
    using SL = SolveLds<SOFT, QCAP, PERSIST, TS, TF>;
    TF *T = (TF *)(B + SL::T);
    auto tc = [&](int j) -> int { return tcol(j) + ((TS < QCAP && j >= TS) ? xo : 0); };   // start of column j (split T: the extension's columns at +xo)
    int *s_meta = (int *)(B + SL::META);
    // per-component slot indices (2 bytes: box slot, position slot; 0xff = none) and the per-step collision-slot mask:
    // the removed slot disappears, higher slots move down by one
    {
        const unsigned b0 = cslot & 0xffu, b1 = (cslot >> 8) & 0xffu;
        const unsigned n0 = (b0 == (unsigned)l) ? 0xffu : ((b0 > (unsigned)l && b0 != 0xffu) ? b0 - 1u : b0);
        const unsigned n1 = (b1 == (unsigned)l) ? 0xffu : ((b1 > (unsigned)l && b1 != 0xffu) ? b1 - 1u : b1);
        cslot = (cslot & 0xffff0000u) | n0 | (n1 << 8);
        if (n0 != b0 && n0 == 0xffu) cslot &= ~0x00030000u;   // the box pair's member bits
        if (n1 != b1 && n1 == 0xffu) cslot &= ~0x000c0000u;   // the position pair's member bits
        const unsigned long long lo = cm & ((1ull << l) - 1ull), hi = (cm >> (l + 1)) << l;
        cm = lo | hi;
    }
    // The rotation of step j is fixed by row l alone: (a_j, b_j) = (current left column, column j+1) at row l, and the left
    // column's entry after the rotation is sqrt(a_j^2 + b_j^2) = a_(j+1).  So a_j^2 is a running sum of squares of ORIGINAL
    // entries of row l -- one wave prefix sum gives every rotation at once (lane j: cosine b_j / |.|, sine a_j / |.|), and
    // the column sweep below carries only two multiply-adds per step on its dependent path (it used to read a_j, b_j out of
    // the updated column and take a reciprocal square root per step: 150+ dependent cycles x the columns right of l, 18 % of a
    // long solve).
    {
        const bool act = lane >= l && lane <= q - 2;
        const double bj = act ? (double)T[tc(lane + 1) + l] : 0.0;
        const double al = (double)T[tc(l) + l];
        double val = bj * bj;
        if (lane == l) val = fma(al, al, val);
        const double Sj = wave_scan_sum(val);   // lane j: a_l^2 + b_l^2 + ... + b_j^2
        B[SL::XS + lane] = Sj;
        LSYNC();
        // a_l is the diagonal entry as it stands (earlier rotations may have left it negative); every later a_j is a norm
        const double Sp = B[SL::XS + (lane > 0 ? lane - 1 : 0)];
        const double aj = (lane == l) ? al : ((Sp > 1e-300) ? Sp * fast_rsq(Sp) : 0.0);
        double cc = 1.0, ss = 0.0;
        if (Sj > 1e-300) { const double inv = fast_rsq(Sj); cc = bj * inv; ss = aj * inv; }
        LSYNC();
        B[SL::RR + 2 * lane] = cc; B[SL::RR + 2 * lane + 1] = ss;   // (cosine, sine) pairs over RR | XS (128 doubles, contiguous): one 16-byte broadcast read each
        LSYNC();
    }
    // the sweep, two columns per round (their LDS reads in flight together); the dependent path is two multiply-adds per column
    static_assert(SL::XS == SL::RR + 64, "rotation pairs span the two staging vectors");
    const double2 *rot = (const double2 *)__builtin_assume_aligned(B + SL::RR, 16);
    double carry = (lane <= l) ? (double)T[tc(l) + lane] : 0.0;
    const int row = lane < l ? lane : lane - 1;
    for (int j0 = l; j0 < q - 1; j0 += 2) {
        const int j1 = j0 + 1 < q - 1 ? j0 + 1 : j0;   // (an odd tail repeats its column: harmless reads, the second update masked)
        const double r0 = (lane <= j0 + 1) ? (double)T[tc(j0 + 1) + lane] : 0.0;
        const double r1 = (lane <= j1 + 1) ? (double)T[tc(j1 + 1) + lane] : 0.0;
        const double2 c0 = rot[j0], c1 = rot[j1];
        const double n0 = c0.x * carry - c0.y * r0;
        carry = c0.y * carry + c0.x * r0;
        if (lane <= j0 + 1 && lane != l) T[tc(j0) + row] = (TF)n0;
        if (j0 + 1 < q - 1) {
            const double n1 = c1.x * carry - c1.y * r1;
            carry = c1.y * carry + c1.x * r1;
            if (lane <= j1 + 1 && lane != l) T[tc(j1) + row] = (TF)n1;
        }
    }
    // the column that dropped out: back to zero up to the end of its group (keeps "zero below the diagonal" for the next append)
    if (lane < ((q + 7) & ~7)) T[tc(q - 1) + lane] = (TF)0.0;
    const bool mv = lane > l && lane < q;
    double v0 = 0, v1 = 0, v2 = 0, ss = 0, d = 0, lam = 0; int meta = 0;
    if (SOFT && lane == l && B[SL::SSS + l] != 0.0) {   // the removed slot leaves its row's map
        const int ml = s_meta[l], tl = (ml >> 8) & 0xff;
        if (tl >= TY_COLL) rmap[(size_t)(tl - TY_COLL) * nrm + (ml >> 16)] = 0xff;
    }
    if (mv) {
        v0 = B[SL::SVEC + 3 * lane]; v1 = B[SL::SVEC + 3 * lane + 1]; v2 = B[SL::SVEC + 3 * lane + 2];
        d = B[SL::SD + lane]; lam = B[SL::SLAM + lane]; meta = s_meta[lane];
        if (SOFT) ss = B[SL::SSS + lane];
    }
    LSYNC();
    if (mv) {
        const int t = lane - 1;
        B[SL::SVEC + 3 * t] = v0; B[SL::SVEC + 3 * t + 1] = v1; B[SL::SVEC + 3 * t + 2] = v2;
        B[SL::SD + t] = d; B[SL::SLAM + t] = lam; s_meta[t] = meta;
        if (SOFT) {
            B[SL::SSS + t] = ss;
            const int tm = (meta >> 8) & 0xff;
            if (ss != 0.0 && tm >= TY_COLL) rmap[(size_t)(tm - TY_COLL) * nrm + (meta >> 16)] = (unsigned char)t;   // (it moved down by one)
        }
    }
    q -= 1;
    LSYNC();
}

// Crash start of the slack variants: append violated acceleration bounds |a| <= alim to the working set WITHOUT a step (lambda = 0; the
// multipliers of the batch are solved afterwards by the refinement pass of the solver).  Two ways, one per call:
//  * from the table.  While the working set holds nothing but bounds appended here and those of an axis are the steps 0 .. m-1 (far
//    from its goal an agent saturates a prefix of the horizon: |a_unc(k)| falls with k), the inverse factor of S = N'H^-1 N is KNOWN:
//    per axis the leading block of Tp = C^-T (H1^-1 = C C', one packed table per cost case, built on the host; a second one for the
//    END of the horizon -- steps 14, 13, .. in falling order, the other common shape), signs sigma_i sigma_j on top, exact zeros between the axes -- H, and with it H^-1, is block diagonal in the three axes (every model matrix is
//    kron(., I3)).  Every violated bound that extends its axis' prefix is appended in ONE pass, its column copied from the table.
//  * by products, one bound PER AXIS and call: the axes are exactly orthogonal in the H^-1 metric, so one pair of triangular products
//    serves three pivots at once (s stacked by the slots' axes; (T's)_j and (T T's)_i only see their own axis) and the factor is bit
//    for bit the one of appending the three one after the other.
// (C4, 10^4 agents: 27 of an agent's 38 "iterations" are such appends -- 17 now come from the table, 10 by products; one bound per
// round they were the largest single piece of the solve launch.)
// Returns the new slot count, the calling lane's updated slot record and flags: bit 0 the factor is still the table's, bit 1 stop the
// crash (dependent pivot: guard only), bit 2 this call used the table, bits 8.. bounds appended (0: nothing was violated).
struct CrashRes { int q; unsigned cslot; int flags; };
template <bool SOFT, int QCAP, bool PERSIST, typename TF = double, int TS = QCAP>
__device__ __attribute__((noinline)) CrashRes crash_append(LdsD *Bl, const LdsD *Gl, const double *tpg, const int lane, const int q, unsigned cslot,
                                                           const double a, const double alim, const double tol, const bool tbl_ok)
{
    using SL = SolveLds<SOFT, QCAP, PERSIST, TS, TF>;
    static_assert(!SOFT || TS == QCAP || TS >= 48, "split T: the crash start (at most 44 slots, then three more per call) stays inside the wave's own columns");
    double *B = (double *)Bl;
    TF *Tf = (TF *)(B + SL::T);
    const double *G = (const double *)Gl;
    int *s_meta = (int *)(B + SL::META);
    const bool comp = lane < N3;
    const int k_l = comp ? lane / 3 : 0, ax_l = comp ? lane - 3 * k_l : 0;
    constexpr int CAP = QCAP - 4 < 44 ? QCAP - 4 : 44;
    constexpr unsigned long long AX0 = 0x0000049249249249ull;   // lanes 0, 3, ..., 42: the 15 components of axis 0
    CrashRes res; res.q = q; res.cslot = cslot; res.flags = tbl_ok ? 1 : 0;
    const unsigned long long vm = __ballot(comp && !(cslot & 0x30000u) && fabs(a) - alim > tol);
    if (vm == 0ull) return res;
    const unsigned long long hm = __ballot(a > 0.0);
    if (tbl_ok) {
        const unsigned long long mem = __ballot(comp && (cslot & 0x30000u) != 0u);
        // per axis: the members are the first m steps of the horizon (dir = 0: table of the rising order) or its LAST m steps (dir = 1:
        // table of the falling order); the run of violated bounds that continues them
        int m0[3], nn[3], dr[3];
        bool pre_ok = true;
        int room = CAP - q;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const unsigned long long axm = AX0 << x, Mx = mem & axm, Vx = vm & axm;
            const int m = __popcll(Mx);
            const bool is_pre = Mx == (axm & ((1ull << (3 * m)) - 1ull));
            const bool is_suf = Mx == (axm & ~((1ull << (3 * (K - m))) - 1ull));
            if (!is_pre && !is_suf) pre_ok = false;
            // rising: first step not in (members | violated); falling: last such step
            const unsigned long long nz = axm & ~(Mx | Vx);
            const int e_up = nz ? ((__ffsll((long long)nz) - 1 - x) / 3) : K;                    // steps 0 .. e_up-1 are members or violated
            const int e_dn = nz ? ((63 - __clzll((long long)nz) - x) / 3) : -1;                  // steps e_dn+1 .. K-1 are members or violated
            const int n_up = is_pre && e_up > m ? e_up - m : 0;
            const int n_dn = is_suf && (K - 1 - e_dn) > m ? (K - 1 - e_dn) - m : 0;
            const int d = (m > 0) ? (is_pre ? 0 : 1) : (n_up > 0 ? 0 : 1);                       // (m == 0: whichever end is violated, the start first)
            int n = d ? n_dn : n_up;
            n = n < room ? n : room;
            room -= n;
            m0[x] = m; nn[x] = n; dr[x] = d;
        }
        const int nb = nn[0] + nn[1] + nn[2];
        if (pre_ok && nb > 0) {
            const int qn = q + nb;
            // (axis, step, sign) of the slot in this lane: old slots from their record, new ones from the runs; ip = position of the
            // step in its axis' order (k rising, K-1-k falling)
            int iax = 0, ik = 0; double isg = 0.0;
            if (lane < q) {
                const int mj = s_meta[lane], cj = mj >> 16;
                ik = cj / 3; iax = cj - 3 * ik;
                isg = (((mj >> 8) & 0xff) == TY_BOXHI) ? 1.0 : -1.0;
            } else if (lane < qn) {
                const int t = lane - q;
                iax = t < nn[0] ? 0 : (t < nn[0] + nn[1] ? 1 : 2);
                const int ta = iax == 0 ? t : (iax == 1 ? t - nn[0] : t - nn[0] - nn[1]);
                const int ma = iax == 0 ? m0[0] : (iax == 1 ? m0[1] : m0[2]);
                const int da = iax == 0 ? dr[0] : (iax == 1 ? dr[1] : dr[2]);
                ik = da ? K - 1 - (ma + ta) : ma + ta;
                isg = ((hm >> (3 * ik + iax)) & 1ull) ? 1.0 : -1.0;
            }
            const int idr = iax == 0 ? dr[0] : (iax == 1 ? dr[1] : dr[2]);
            const int ip = idr ? K - 1 - ik : ik;
            const double *tp = B + SL::RR + ip * (31 - ip) / 2 - ip;   // tp[pj] = Tp(ip, pj), pj >= ip
            // the case's packed tables through the two staging vectors (120 of their 128 doubles), one order at a time
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                if (!((nn[0] && dr[0] == pass) || (nn[1] && dr[1] == pass) || (nn[2] && dr[2] == pass))) continue;
                {
                    const double *tg = tpg + pass * TAB_TP_CASE;
                    const double t0 = tg[lane], t1 = tg[lane + 64 < TAB_TP_CASE ? lane + 64 : 0];
                    LSYNC();
                    B[SL::RR + lane] = t0;
                    if (lane + 64 < TAB_TP_CASE) B[SL::RR + 64 + lane] = t1;
                    LSYNC();
                }
                int j = q;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    if (dr[x] != pass) { j += nn[x]; continue; }
                    for (int u = 0; u < nn[x]; ++u, ++j) {
                        const int pj = m0[x] + u, kj = dr[x] ? K - 1 - pj : pj;
                        const double sgj = ((hm >> (3 * kj + x)) & 1ull) ? 1.0 : -1.0;
                        if (lane < ((j + 8) & ~7)) {
                            const bool nzr = lane <= j && lane < qn && iax == x;   // (rows of the axis: their positions are <= pj)
                            Tf[tcol(j) + lane] = (TF)(nzr ? (isg * sgj) * tp[pj] : 0.0);
                        }
                    }
                }
            }
            if (lane >= q && lane < qn) {
                B[SL::SVEC + 3 * lane] = iax == 0 ? isg : 0.0; B[SL::SVEC + 3 * lane + 1] = iax == 1 ? isg : 0.0; B[SL::SVEC + 3 * lane + 2] = iax == 2 ? isg : 0.0;
                B[SL::SD + lane] = alim; B[SL::SLAM + lane] = 0.0; B[SL::SSS + lane] = 0.0;
                s_meta[lane] = ik | ((isg > 0.0 ? TY_BOXHI : TY_BOXLO) << 8) | ((3 * ik + iax) << 16);
            }
            if (comp) {   // the component's own slot index
                const int mx = ax_l == 0 ? m0[0] : (ax_l == 1 ? m0[1] : m0[2]), nx = ax_l == 0 ? nn[0] : (ax_l == 1 ? nn[1] : nn[2]);
                const int dx = ax_l == 0 ? dr[0] : (ax_l == 1 ? dr[1] : dr[2]);
                const int off = ax_l == 0 ? 0 : (ax_l == 1 ? nn[0] : nn[0] + nn[1]);
                const int pl = dx ? K - 1 - k_l : k_l;
                if (pl >= mx && pl < mx + nx)
                    cslot = (cslot & ~0xffu) | (unsigned)(q + off + pl - mx) | (a > 0.0 ? 0x10000u : 0x20000u);
            }
            LSYNC();
            res.q = qn; res.cslot = cslot; res.flags = 1 | 4 | (nb << 8);
            return res;
        }
    }
    // by products: the lowest violated component of every axis
    int pc[3], pk[3]; double psg[3], spa[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const unsigned long long mx = vm & (AX0 << x);
        pc[x] = mx ? (__ffsll((long long)mx) - 1) : -1;
        pk[x] = pc[x] >= 0 ? pc[x] / 3 : 0;
        psg[x] = (pc[x] >= 0 && ((hm >> pc[x]) & 1ull)) ? 1.0 : -1.0;
        spa[x] = G[pk[x] * 31];
    }
    // s = N_W' H^-1 [n_x n_y n_z] stacked: slot j takes the entry of its own axis' pivot
    int jax = 0; double sv = 0.0;
    if (lane < q) {
        const int mj = s_meta[lane];
        const int cj = mj >> 16, gj = mj & 0xff;
        jax = cj - 3 * (cj / 3);
        const double sgj = (((mj >> 8) & 0xff) == TY_BOXHI) ? 1.0 : -1.0;
        const int pkj = jax == 0 ? pk[0] : (jax == 1 ? pk[1] : pk[2]);
        const bool has = (jax == 0 ? pc[0] : (jax == 1 ? pc[1] : pc[2])) >= 0;
        const double sgp = jax == 0 ? psg[0] : (jax == 1 ? psg[1] : psg[2]);
        sv = has ? G[gj * 30 + pkj] * (sgj * sgp) : 0.0;
    }
    B[SL::XS + lane] = sv; LSYNC();
    const double dvj = t_tmul2<QCAP, SL::T, SL::XS, TS, TF, SOFT>(B, lane, q);
    B[SL::RR + lane] = dvj; LSYNC();
    const double ri = t_mul2<QCAP, SL::T, SL::RR, TS, TF, SOFT>(B, lane, q);
    LSYNC();
    const double d2 = (lane < q) ? dvj * dvj : 0.0;
    double irho[3]; bool okx[3];
    bool stop = false;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const double dlt = spa[x] - wave_sum0(jax == x ? d2 : 0.0);
        okx[x] = pc[x] >= 0 && dlt > 1e-9 * spa[x];        // (distinct bounds are independent; guard only)
        if (pc[x] >= 0 && !okx[x]) stop = true;
        irho[x] = okx[x] ? fast_rsq(dlt) : 0.0;
    }
    // the new columns [-r/rho ; 1/rho] (own axis; exact zeros elsewhere), slots, per-component slot index
    int qn = q;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        if (!okx[x]) continue;
        if (lane < ((qn + 8) & ~7))
            Tf[tcol(qn) + lane] = (TF)((lane < q) ? ((jax == x) ? (-ri * irho[x]) : 0.0) : ((lane == qn) ? irho[x] : 0.0));
        if (lane == 0) {
            B[SL::SVEC + 3 * qn] = x == 0 ? psg[x] : 0.0; B[SL::SVEC + 3 * qn + 1] = x == 1 ? psg[x] : 0.0; B[SL::SVEC + 3 * qn + 2] = x == 2 ? psg[x] : 0.0;
            B[SL::SD + qn] = alim; B[SL::SLAM + qn] = 0.0; B[SL::SSS + qn] = 0.0;
            s_meta[qn] = pk[x] | ((psg[x] > 0.0 ? TY_BOXHI : TY_BOXLO) << 8) | (pc[x] << 16);
        }
        if (lane == pc[x]) cslot = (cslot & ~0xffu) | (unsigned)qn | (psg[x] > 0.0 ? 0x10000u : 0x20000u);
        ++qn;
    }
    LSYNC();
    res.q = qn; res.cslot = cslot; res.flags = (stop ? 2 : 0) | ((qn - q) << 8);   // (the slots of an axis are no prefix in step order any more)
    return res;
}

// Closed loops, tiny launches: the work of post_step_kernel for one agent (dmpc_soft_bound.m:132-134, the history column,
// ReachedGoal.m:3-11), done by the wave that produced the agent's step; p_out, v_out, a_out: lanes 0..2 hold the first horizon column.
// The scene's maximum / OR / count are order-independent, so which wave finishes last changes nothing.
__device__ __forceinline__ void post_step_part(const KargPtr Qp, const int lane, const int gid, const int scene, const bool solved,
                                               const int status, const double p_out, const double v_out, const double a_out)
{
    {
        double xn = 0.0, vn = 0.0, an = 0.0, e2 = 0.0;
        if (lane < 3) {
            const size_t b = (size_t)gid * 3 + lane;
            xn = solved ? p_out : Qp->post_xp[b]; vn = solved ? v_out : Qp->post_xv[b]; an = solved ? a_out : Qp->post_xa[b];
            if (solved) { Qp->post_xp[b] = xn; Qp->post_xv[b] = vn; Qp->post_xa[b] = an; }
            const size_t ho = ((size_t)gid * Qp->post_KT + Qp->post_k) * 3 + lane;
            Qp->post_pk[ho] = xn; Qp->post_vk[ho] = vn; Qp->post_ak[ho] = an;
            const double dd = xn - Qp->pf[b];
            e2 = dd * dd;
        }
        const double dx2 = readlane_d(e2, 0), dy2 = readlane_d(e2, 1), dz2 = readlane_d(e2, 2);
        if (lane == 0) {
            const double dist = sqrt(dx2 + dy2 + dz2);
            atomicMax(Qp->post_max + scene, (unsigned long long)__double_as_longlong(dist));
            atomicOr(Qp->post_or + scene, status);
            __threadfence();
            if (atomicAdd(Qp->post_cnt + scene, 1) == Qp->c_count - 1) {   // the scene's last agent of this step
                const unsigned long long mb = atomicExch(Qp->post_max + scene, 0ull);
                const int orv = atomicExch(Qp->post_or + scene, 0);
                Qp->post_cnt[scene] = 0;
                const int reached = __longlong_as_double((long long)mb) < Qp->post_tol ? 1 : 0;
                Qp->post_flags[(size_t)scene * 2] = reached; Qp->post_flags[(size_t)scene * 2 + 1] = orv;
                if (Qp->post_done && (reached || (orv & ~ST_SOLVED))) Qp->post_done[scene] = 1;
            }
        }
    }
}

// `bidx`: the workgroup's index (one-agent-per-workgroup launches, renumbered XCD-aware and sent through the launch order
// here) or -- persistent form -- the AGENT the wave is about to solve (the queue position already resolved through the order
// by the persistent loop); `smem`: this wave's LDS; `shtab`: the workgroup-shared tables (persistent
// form).
template <bool SOFT, int QCAP, bool PERSIST, int TS = QCAP, typename TF = double>
__device__ __forceinline__ void solve_body(const StepParams &P, const int lane, const int bidx, const int nblocks,
                                           unsigned char *smem, const double *shtab, const bool want_ticket, int &ticket, bool &claimed,
                                           const int ext0 = 0 /* split T: doubles from this wave's block to extension 0 of the workgroup's pool */)
{
    // Persistent form: the wave's NEXT queue ticket is claimed (lane 0, result left in flight in `ticket`) when the agent is as good as done
    // -- the first violation scan that finds nothing, or the start of the output stage -- so that the atomic's latency hides behind the
    // output stage and nothing is claimed ahead of a solve: a position claimed before a 300-500 us infeasibility proof waited behind
    // it, and such parked agents were the last to end the launch (round 3: the waves ended 779-878 us, busy fraction 0.90).
#define CLAIM_NEXT() do { if (PERSIST && want_ticket && !claimed) { claimed = true; \
        if (lane == 0) ticket = atomicAdd(kernarg_params()->counter, 1); } } while (0)
    using SL = SolveLds<SOFT, QCAP, PERSIST, TS, TF>;
    constexpr bool soft = SOFT;
    constexpr bool F32T = sizeof(TF) == 4;
    static_assert(!F32T || TS == QCAP, "the fp32 factor is not split");
    // fp32 factor: r = T T's carries ~1e-7 relative, so a DEPENDENT pivot shows a residual nu'H^-1 nu of ~1e-13 s_pp instead of ~1e-30:
    // the dependence threshold moves up, and the refinement gets more passes (it contracts by ~1e-7 cond(S) per pass instead of at once)
    const double DEP_TOL = F32T ? P.dep_tol_f32 : 1e-13;
    constexpr int REFINE_PASSES = F32T ? 10 : 3;
    // (round 5: also the slack kernels of large scenes -- QCAP 56 with 48 own columns: seven waves per CU instead of five one-agent workgroups)
    static_assert(TS == QCAP || (PERSIST && TS % 8 == 0 && TS < QCAP && (!SOFT || TS >= 48)), "split T: persistent kernels");
    const int nrmax = P.nrmax, var = P.variant;
    // split T: offset (doubles from the wave's base) that puts column j >= TS of the factor at tcol(j) + xo; 0: no extension held.
    // The pool: a bit mask of the free extensions behind the workgroup's tables; lane 0 takes the lowest free one (LDS atomic), waits
    // if there is none (the holders never wait for anything: they finish), and the agent gives it back -- zeroed -- when it is done.
    int xo = 0;
#define ENSURE_EXT(qn_) do { if (TS < QCAP && (qn_) >= TS && xo == 0) { \
        unsigned *pool__ = (unsigned *)(shtab) + (PERSIST_TABLE_BYTES - 16) / 4; \
        int sl__ = -1; \
        if (lane == 0) { \
            for (;;) { \
                const unsigned m__ = __atomic_load_n(pool__, __ATOMIC_RELAXED); \
                if (m__) { const int b__ = __ffs((int)m__) - 1; if ((atomicAnd(pool__, ~(1u << b__)) >> b__) & 1u) { sl__ = b__; break; } } \
                else __builtin_amdgcn_s_sleep(16); \
            } \
        } \
        sl__ = __builtin_amdgcn_readfirstlane(sl__); \
        xo = ext0 + sl__ * ext_doubles(QCAP, TS) - tcol(TS); \
    } } while (0)
    int vb = bidx;
    if (!PERSIST) {   // XCD-aware renumbering (see step_body)
        const int nb = nblocks, x = bidx & 7, y = bidx >> 3;
        int off = 0;
        for (int xx = 0; xx < x; ++xx) off += (nb - xx + 7) >> 3;
        vb = off + y;
        if (P.order) vb = P.order[bidx];   // heaviest agents first (order_kernel) / tier-2 list
    }
    const int scene = vb / P.c_count, ci = vb - scene * P.c_count;
    const int cl = P.c_first + ci;
    const int gid = scene * P.c_count + ci;

    double *B = (double *)__builtin_assume_aligned(smem, 16);
    double *r_eps = B + SL::VAR;                                                   // soft variants: nrmax doubles
    unsigned char *r_fl = (unsigned char *)(B + SL::VAR + (soft ? nrmax : 0));     // soft variants: nrmax bytes of flags; slack-free: nrmax BITS
    unsigned *r_bits = (unsigned *)r_fl;   // (their only flag is RF_COLL; 80 instead of 640 bytes per wave: a ninth persistent wave fits the CU's LDS)
    // Slack kernels, round 4: per row the working-set slot of its collision constraint, of its pin eps <= 0 and of its bound eps >= slb
    // (0xff: none; only slots with a slack coefficient are entered).  The slack part of the residual used to be found by every lane
    // walking ALL slack-carrying slots of the working set one after the other (20-30 dependent LDS round trips per iteration for an
    // agent of the 10^4-agent scene: a quarter of its iteration); with the maps a slot's lane reads the up to three slots of its own row.
    unsigned char *m_row = r_fl + nrmax;   // [3][nrmax]
    auto row_slots = [&](int row, int &s0, int &s1, int &s2) {   // ascending (the order of the sums they replace), 0xff last
        const int a = m_row[row], b = m_row[nrmax + row], c = m_row[2 * nrmax + row];
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        s0 = lo < c ? lo : c; s2 = hi > c ? hi : c; s1 = lo > c ? lo : (hi < c ? hi : c);
    };
    int *s_meta = (int *)(B + SL::META);
    // collision rows: per-agent slice of the global scratch written by the scan kernel (lane = row: coalesced)
    const size_t per = (size_t)nrmax * (soft ? 7 : 4);
    double *g_rows = P.rowbuf + (size_t)gid * per;
    double *r_xi = g_rows, *r_b = g_rows + 3 * (size_t)nrmax;
    double *r_sd = soft ? r_b + nrmax : nullptr, *r_st = soft ? r_b + 2 * (size_t)nrmax : nullptr, *r_slb = soft ? r_b + 3 * (size_t)nrmax : nullptr;
    int *r_kc = P.rowkc + (size_t)gid * nrmax;
    int *hdr = P.hdr + (size_t)gid * 8;

    // ---------------------------------------------------------------- the global loads of the set-up in two rounds
    // (hand-off header + agent state, then the register-cached rows, which need the row count: two memory round trips
    // instead of a chain of four -- header flag, state, row count, rows.  Rows addressed without the count, i.e. reads of
    // scratch the scan never wrote, were measured slower: those lines come cold from HBM and the whole set-up waits for them.)
    struct { int x, y, z, w; } h0, h1;   // the scan's hand-off header: 8 ints
    // (readfirstlane: the header is read with vector loads -- the scan wrote it in this launch's lifetime, no scalar load -- and everything
    // derived from it, the row count, the status word, the ladder start, would be compiled as lane-dependent: masked loops, vector compares)
    h0.x = UNI(hdr[0]); h0.y = UNI(hdr[1]); h0.z = UNI(hdr[2]); h0.w = UNI(hdr[3]); h1.x = UNI(hdr[4]); h1.y = UNI(hdr[5]); h1.z = UNI(hdr[6]); h1.w = UNI(hdr[7]);
    const int stq = P.only_flagged ? UNI(P.status[gid]) : ST_QOVER;
    double po[3], vo[3], ao[3], pf[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        po[d] = P.x_p[3 * gid + d]; vo[d] = P.x_v[3 * gid + d];
        ao[d] = P.x_a[3 * gid + d]; pf[d] = P.pf[3 * gid + d];
    }
    if (P.only_flagged && !(stq & ST_QOVER)) return;   // tier 2: only agents that overflowed tier 1
    if (h1.x & 8) return;                              // agent of a scene that already stopped
    if (h1.x & 16) {   // finished by the scan (unconstrained exit): nothing to solve; a fused post-step still takes its outputs
        if (P.post_on) {
            const KargPtr Qp = kernarg_params();
            const int st_done = Qp->status[gid];
            double p1 = 0.0, v1 = 0.0, a1 = 0.0;
            if (lane < 3) { p1 = Qp->p_out[(size_t)gid * N3 + lane]; v1 = Qp->v_out[(size_t)gid * N3 + lane]; a1 = Qp->a_out[(size_t)gid * N3 + lane]; }
            post_step_part(Qp, lane, gid, scene, (st_done & ST_SOLVED) != 0, st_done, p1, v1, a1);
        }
        return;
    }

    // the scan's branch record
    int nr = h0.x, status = h0.w;
    const int nrows_built = h0.y, viol_k = h0.z;
    const bool violation = (h1.x & 1) != 0, rows_exist = h1.y != 0;
    const bool cppv = (var == VAR_CPP || var == VAR_CPP2);
#ifndef DMPC_HARD_RC
#define DMPC_HARD_RC 2
#endif
    constexpr int RC = soft ? 1 : DMPC_HARD_RC;   // register cache of the first collision rows (RC per lane; the rest is streamed from the L2-resident scratch)
    double rcx0[2], rcx1[2], rcx2[2], rcb[2], rcsd[2], rcslb[2], rcst[2];   // (rcst: the slack's linear cost -- read by every re-derivation of the iterate, a global round trip each until round 5)
    float rcw[2];
    int rckc[2];
    rcx0[1] = rcx1[1] = rcx2[1] = rcb[1] = rcsd[1] = rcslb[1] = rcst[1] = 0.0; rcw[1] = 0.f; rckc[1] = 0;
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        const int i = lane + 64 * c;
        const int ii = i < nr ? i : 0;
        rcx0[c] = r_xi[3 * ii]; rcx1[c] = r_xi[3 * ii + 1]; rcx2[c] = r_xi[3 * ii + 2];
        rcb[c] = r_b[ii]; rckc[c] = r_kc[ii];
        rcsd[c] = soft ? r_sd[ii] : 0.0; rcslb[c] = soft ? r_slb[ii] : 0.0; rcst[c] = soft ? r_st[ii] : 0.0;
    }

    // ---------------------------------------------------------------- cost case + tables (a7, :43-58)
    const int ccase = UNI(cost_case(var, po[0] - pf[0], po[1] - pf[1], po[2] - pf[2], rows_exist));   // (the state comes through vector loads: the table base of every Gram lookup would be per-lane arithmetic)
    const double qw = ccase == 0 ? P.Qfar : (ccase == 1 ? P.Qnear : P.Q1);
    const double sw = ccase == 2 ? ((var == VAR_ALL3) ? 10.0 : P.S1) : P.Sfree;
    const double *G, *Lt;
    if (PERSIST) { G = shtab + ccase * TAB_CASE_DOUBLES; Lt = shtab + 3 * TAB_CASE_DOUBLES; }
    else {
        const double *src = P.tables + (size_t)ccase * TAB_CASE_DOUBLES, *srcl = P.tables + 3 * TAB_CASE_DOUBLES;
        for (int i = lane; i < TAB_CASE_DOUBLES; i += 64) B[SL::TAB + i] = src[i];
        for (int i = lane; i < TAB_L_DOUBLES; i += 64) B[SL::TAB + TAB_CASE_DOUBLES + i] = srcl[i];
        G = B + SL::TAB; Lt = G + TAB_CASE_DOUBLES;
    }
    // Every LDS value the solver can touch must be finite: the matrix-vector products read whole groups of 8 unmasked
    // (stale columns only ever meet x_j = 0).  The region starts at zero for every agent, so a non-finite input of one
    // agent cannot reach the next agent of a persistent wave.
    for (int i = lane; i < SL::TAB; i += 64) B[i] = 0.0;
    LSYNC();

    const int k_l = lane < N3 ? lane / 3 : 0, ax_l = lane < N3 ? lane - 3 * k_l : 0;
    const bool comp = lane < N3;
    // unconstrained minimiser per axis: a_unc = -H1^-1 f,  f = -2(q L_K'(pf - A0_K x0) + s [ao;0..])
    //   => a_unc(k) = 2 q g (H1^-1 L')[k][K-1] + 2 s ao H1^-1[k][0],  g = pf - (po + K h vo)      (:88/:93)
    double a_unc = 0.0, w_unc = 0.0;   // (w_unc: a register of the slack variants; the slack-free ones keep it in LDS, B[SL::WU])
    const double gax = comp ? goal_gap(sel3(pf, ax_l), sel3(po, ax_l), sel3(vo, ax_l), P.h) : 0.0;
    const double ao_l = comp ? sel3(ao, ax_l) : 0.0;
    if (comp) {
        a_unc = unc_entry(qw, sw, gax, ao_l, G[k_l * 30 + 15 + (K - 1)], G[k_l * 30]);
        // w_unc = Lambda a_unc from the same table
        w_unc = unc_entry(qw, sw, gax, ao_l, G[(15 + k_l) * 30 + 15 + (K - 1)], G[(15 + k_l) * 30]);
        if (!SOFT) B[SL::WU + lane] = w_unc;
    }

    // Dual-bound certificate (slack-free variants): the iterate of the dual method minimises the cost over its working
    // set, so its cost value `dual` is a lower bound of the constrained optimum, rising with every step by
    // t delta (lambda_p + t/2).  Every feasible point lies in the box |a| <= alim, where the cost is at most
    //   fbound = 3/2 alim^2 sum|H1(i,j)| + alim sum|f_i|; once `dual` exceeds that the QP is infeasible.
    double dual = 0.0, fbound = INFINITY;
    if (!soft) {
        double f_l = 0.0;
        if (comp) {
            const double LKk = 0.5 * P.h * P.h + (double)(K - 1 - k_l) * P.h * P.h;   // Lambda(K, k)
            f_l = -2.0 * qw * LKk * gax - ((k_l == 0) ? 2.0 * sw * ao_l : 0.0);
        }
        dual = 0.5 * wave_sum0(f_l * a_unc);
        const double fabs_sum = wave_sum0(fabs(f_l));
        fbound = 1.5 * P.alim * P.alim * P.hsum[ccase] + P.alim * fabs_sum;
        fbound += 1e-6 * (fabs(fbound) + fabs(dual));
    }

    // per-lane constants of component (k_l, ax_l)
    double whi_l = 0.0, wlo_l = 0.0;
    float wbox_f = 0.f, wpos_f = 0.f;
    if (comp) {
        const double sh = (double)(k_l + 1) * P.h * sel3(vo, ax_l);   // A_initp(k,:) [po;vo] - po
        whi_l = sel3(P.pmax, ax_l) - sel3(po, ax_l) - sh;        // pmax - A0 x0  (:72)
        wlo_l = sel3(P.pmin, ax_l) - sel3(po, ax_l) - sh;
        // Pivot weights (fp32: the choice only orders the pivots; the minimiser does not depend on it).  Slack-free variants: the entering
        // constraint is the one farthest from feasibility in the metric of the problem, violation / |n|_{H^-1} -- with the collision rows
        // of solveHardDMPC EIGHT times as heavy (below).  Slack variants: plain violation for the acceleration bounds and the workspace
        // walls, 4 violation / |xi| for the collision rows.  Both say "rows first": a bound that is violated now is often not at the
        // optimum once the rows have moved the iterate, and every such detour costs an append and a drop.  Measured in round 3 (the
        // rule had been violation / |n|_{H^-1} for everything): solveSoftDMPCbound replay mean 1.10 -> 1.01 iterations, longest agent
        // 115 -> 67, solve launch 0.76 -> 0.48 ms; the 10^4-agent scene 1.16 -> 0.95-1.0 ms; solveHardDMPC 7.01 -> 6.52 iterations.
        // (A slack row brings its pin along -- two slots -- so "rows first" passes through larger working sets: where that runs out of the
        // 64 slots of the last tier the level is solved again with the metric rule, `alt_rule`, which adds the rows as sparingly as the
        // bounds: three agent-steps of solveSoftDMPCall in the recorded scenes of the randomized campaign.)
        if (SOFT) { wbox_f = 1.f; wpos_f = 1.f; }
        else { wbox_f = __builtin_amdgcn_rsqf((float)G[k_l * 31]); wpos_f = __builtin_amdgcn_rsqf((float)G[(15 + k_l) * 31]); }
    }
    const float rowmul = (var == VAR_HARD) ? 8.f : 1.f;   // (solveHardDMPCOnDemand, solveEllipDMPC: measured neutral to slightly worse with heavier rows)
    bool alt_rule = false;   // slack variants: the metric rule for everything (second attempt of a level that ran out of slots)
#ifdef DMPC_PIVOT_EXPLORE
    const float explore_row = ((P.pivot_explore & 15) ? ldexpf(1.f, (P.pivot_explore & 15) - 8) : 1.f);
#else
    constexpr float explore_row = 1.f;
#endif
    auto row_weight = [&](double x0, double x1, double x2, int kc, double sd) -> float {
        if (SOFT) {
            if (alt_rule) return __builtin_amdgcn_rsqf((float)(G[(15 + kc) * 31] * (x0 * x0 + x1 * x1 + x2 * x2) + 0.5 * sd * sd));
            return explore_row * 4.f * __builtin_amdgcn_rsqf((float)(x0 * x0 + x1 * x1 + x2 * x2));
        }
        return explore_row * rowmul * __builtin_amdgcn_rsqf((float)(G[(15 + kc) * 31] * (x0 * x0 + x1 * x1 + x2 * x2)));
    };
#pragma unroll
    for (int c = 0; c < RC; ++c) rcw[c] = row_weight(rcx0[c], rcx1[c], rcx2[c], rckc[c], rcsd[c]);
    float wslk_u = 1.4142135f, wslk_l = 1.4142135f;   // the slack bounds' weights (|n|^2 = 1/2)
#ifdef DMPC_PIVOT_EXPLORE   // development build: extra multipliers 2^(field - 8) packed into P.pivot_explore (0 = 1): bits 0-3 rows, 4-7 eps <= 0, 8-11 eps >= slb, 12-15 walls, 16-19 bounds
    {
        const int pe = P.pivot_explore;
        auto mul = [&](int sh) -> float { const int f = (pe >> sh) & 15; return f ? ldexpf(1.f, f - 8) : 1.f; };
        wslk_u *= mul(4); wslk_l *= mul(8); wpos_f *= mul(12); wbox_f *= mul(16);
    }
#endif

    // ---------------------------------------------------------------- a7: dual active-set solve
    const bool ladder = soft && (var == VAR_BOUND || var == VAR_BOUND2 || var == VAR_ALL3 || cppv);
#ifndef DMPC_LADDER_CERT_AFTER
#define DMPC_LADDER_CERT_AFTER 8    // (bound replay, 512 scenes: 16: 0.994 ms per step, 12: 0.980, 8: 0.965, 4: 0.959 -- the certificate runs for every agent that gets this far)
#endif
    constexpr int LADDER_CERT_AFTER = DMPC_LADDER_CERT_AFTER;
#ifndef DMPC_FARKAS_AFTER
#define DMPC_FARKAS_AFTER 8
#endif
    constexpr int FARKAS_AFTER = DMPC_FARKAS_AFTER;
    const int max_tries = P.max_tries > 0 ? P.max_tries : (cppv ? 21 : 30);
    // hdr[6] = retry-ladder levels the scan certified infeasible (the ladder starts behind them, the skipped tries counted).
    // A first-tier launch that runs out of working-set slots hands the agent over UNTOUCHED (its row scalings undone, nothing
    // recorded): the second tier repeats the whole solve, on the path an uninterrupted solve takes -- the result of an agent
    // must not depend on how deep the launch was that it ran in.
    int tries = h1.z, iters_total = 0, maxq = 0, q = 0;
    int cost = 0;   // work estimate in quarter microseconds (wave-uniform, scalar registers): the next step's launch-order key (P.cost_out)
    int scale_pow = 0;   // the rows' slack bound and penalty currently carry the factor 2^scale_pow
#ifdef DMPC_DEV_TRACE
    int dev_nfast = 0, dev_rounds = 0, dev_negdrops = 0, dev_tbl = 0, dev_gen = 0, dev_try = 0;
    // development: cycles of the traced agent by phase (s_memtime): 0 pivot scan, 1 pivot descriptor + pin, 2 s / T's / T T's,
    // 3 residual + direction + delta, 4 ratio test + step + append, 5 drops, 6 verification / refinement, 7 ladder certificate +
    // ladder step, 8 set-up; counts: 10 verifications, 11 drops, 12 certificate calls
    long long phv[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 13 s, 14 T's, 15 ratio test, 16 step + certificates, (4: append)
    long long ph_last = __builtin_amdgcn_s_memtime();
    const bool ph_on = P.dbg && gid == P.dbg_agent;
#define PH(i_) do { const long long t__ = __builtin_amdgcn_s_memtime(); if (ph_on) phv[i_] += t__ - ph_last; ph_last = t__; } while (0)
#define PHC(i_) do { if (ph_on) phv[i_] += 1; } while (0)
#elif defined(DMPC_ISA_MARK)   // development (tools/isa_stats.sh): phase boundaries as comments in the instruction stream
#define PH(i_) asm volatile("; @@PH " #i_ ::: "memory")
#define PHC(i_) do { } while (0)
#else
#define PH(i_) do { } while (0)
#define PHC(i_) do { } while (0)
#endif
    bool solved = false;
    double a = 0.0, w = 0.0;
    const double tol = 1e-10;

    if (status & ST_INFEAS) tries = 1;   // certified infeasible by the scan (single attempt: hard rows only)
    if (soft && tries > 0 && !(status & ST_INFEAS)) {
        if (tries >= max_tries) { status |= ST_INFEAS; tries = max_tries; }
        else {
            const double f = ldexp(1.0, tries);
            scale_pow = tries;
            for (int i = lane; i < nr; i += 64) { r_slb[i] *= f; r_st[i] *= f; }
            rcslb[0] *= f; rcslb[1] *= f; rcst[0] *= f; rcst[1] *= f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    if (!(status & (ST_COLL | ST_CAPACITY | ST_INFEAS))) {
        // cslot: byte 0 = slot of the component's box constraint (0xff none), byte 1 = slot of its workspace bound,
        // bits 16..19 = membership of BOXHI | BOXLO | POSHI | POSLO; cm: collision slots that constrain step k_l
        unsigned cslot = 0x0000ffffu;
        unsigned long long cm = 0ull;
        // working-set membership of the collision rows: slack-free variants keep the bits of the register-cached rows in
        // a register of the owning lane (bit c = row lane + 64 c), the rest (and all flags of the soft variants) in LDS bytes
        unsigned rcfl = 0;
        int nlive = 0;
        // Every ladder level starts from scratch.  (A warm ladder -- the next level keeping working set, factor and multipliers of the
        // failed one: a ladder step changes right-hand sides and the slacks' linear cost, no constraint normal -- was built in round 2 and
        // is gone: an infeasible try often ends in a nearly degenerate working set, and the randomized campaign found 6 of 470 550
        // agent-steps that ended infeasible or with a wrong retry count on the factor it leaves behind, even with a conditioning guard.)
        constexpr bool warm = false;
        bool cert_known = false;   // the level about to start has passed the ladder certificate (it is not certainly infeasible)
        PH(8);
        // solveSoftDMPCall (rows on three horizon steps): FEASIBILITY OF A LADDER LEVEL FIRST (round 4).  A level is feasible iff the rows hold
        // with every slack at its bound, eps = slb 2^t -- a slack-free problem (no slack variables, no pins: a third of the constraints) on
        // which the dual method proves infeasibility in tens of iterations, where the soft problem of the same level needed ~250 (levels
        // that are jointly but not per-step infeasible: the per-step ladder certificate does not see them; 700 of 51 200 agents of the
        // replay took more than 100 iterations, up to 1 247 over 4-5 levels: 12.8 ms per step against 0.45-0.7 ms for every other variant).
        // `hp`: this pass of the loop is that check -- rows as hard constraints with right-hand side b - d slb, nothing recorded; an
        // infeasible verdict takes the ladder step exactly as a failed try would, a feasible one is followed by the real solve of the level.
        const bool hp_variant = SOFT && var == VAR_ALL3 && violation && !P.no_level_check;
        int lev_skip = 0;   // ladder levels above the current one that the last infeasibility proof covers as well
        bool hp = false, level_checked = false;
        while (tries < max_tries) {
            tries++;
            hp = hp_variant && !level_checked;
            int rc = 0;   // 0 running/ok, 1 infeasible, 2 capacity, 3 itercap
            int iters = 0;
            if (!warm) {
                q = 0; cslot = 0x0000ffffu; cm = 0ull; rcfl = 0; nlive = 0;
                if (soft) { for (int i = lane; i < nr; i += 64) { r_fl[i] = 0; m_row[i] = 0xff; m_row[nrmax + i] = 0xff; m_row[2 * nrmax + i] = 0xff; } }
                else { for (int i = lane; i < ((nr + 31) >> 5); i += 64) r_bits[i] = 0u; }
                a = a_unc; w = SOFT ? w_unc : (comp ? B[SL::WU + lane] : 0.0);
                if (comp) { B[SL::A + lane] = a; B[SL::W + lane] = w; }
                if (soft) for (int i = lane; i < nr; i += 64) r_eps[i] = 0.0;
            }
            double g_l = 0.0;   // gradient of the cost at the iterate (slack-free variants): Farkas test against the box
            // Crash start of the acceleration bounds.  Far from its goal an agent saturates most of its 45 bounds |a| <= alim, and
            // the dual method would add them one full iteration each.  While `crash` is on, only those bounds are candidates and
            // a candidate is APPENDED to the factor without a step (no residual, no direction, no ratio test: a third of an
            // iteration); when no further bound is violated at the stale iterate the multipliers of the whole working set are
            // solved at once (lambda += T T' rho, the refinement pass) and the primal re-derived.  Multipliers that come out
            // negative are dropped and the crash ends; what it leaves is a working set with lambda >= 0 whose equality-constrained
            // minimiser is the iterate -- a valid state of the dual method, which continues from there with every constraint.
            // (slack-free variants: measured on solveHardDMPC, where rows sit on every horizon step, the guess is poor and the crash costs
            // 16 % -- compiled out there)
            bool crash = SOFT && (warm || (P.crash_min > 0 && __popcll(__ballot(comp && fabs(a_unc) - P.alim > tol)) >= P.crash_min));
            bool crash_stop = warm;          // warm: straight to the finish (solve the multipliers of the kept working set)
            bool crash_box = !warm;          // every slot of the batch is an acceleration bound
            int crash_rounds = 0, nfast = 0, accept_drops = 0;
            bool pristine = !warm;
            const double dual0 = dual;
            LSYNC();

            // nu = sgn * n_p - N_W r assembled as one vector in a-space, from the vector r in B[RR]:
            //   nu_a = U + Lambda' Y   (U: box part, Y: position/collision part)
            // returns nu for this lane's component; Y is staged in B[Y], nu is left in B[NU]
            auto residual = [&](const double pU, const double pY) -> double {
                double nu = 0.0;
                if (comp) {
                    const unsigned sb = cslot & 0xffu, sp = (cslot >> 8) & 0xffu;
                    const double rb = B[SL::RR + (sb & 63u)], rp = B[SL::RR + (sp & 63u)];
                    // BOXHI: vec = +e, BOXLO: vec = -e  (nu -= r * vec): sign from the member bits, 0 without a slot
                    const double fb = (cslot & 0x10000u) ? -1.0 : ((cslot & 0x20000u) ? 1.0 : 0.0);
                    const double fp = (cslot & 0x40000u) ? -1.0 : ((cslot & 0x80000u) ? 1.0 : 0.0);
                    const double U = fma(fb, rb, pU);
                    double Y = fma(fp, rp, pY);
                    // (slack variants: four collision slots per round, the eight LDS reads of a round in flight together -- solveSoftDMPCbound
                    // puts all rows on ONE horizon step, whose three lanes then walk every collision slot of the working set here; the
                    // slack-free variants spread their rows over the steps, a lane meets one or two slots: one per round)
                    constexpr int MW = SOFT ? 4 : 1;
                    unsigned long long m = cm;
                    while (__any(m != 0ull)) {
                        int jj[MW]; bool hv[MW]; double rj[MW], vj[MW];
#pragma unroll
                        for (int u = 0; u < MW; ++u) {
                            hv[u] = m != 0ull;
                            jj[u] = hv[u] ? (__ffsll((long long)m) - 1) : 0;
                            m &= m - 1ull;
                        }
#pragma unroll
                        for (int u = 0; u < MW; ++u) { rj[u] = B[SL::RR + jj[u]]; vj[u] = B[SL::SVEC + 3 * jj[u] + ax_l]; }
#pragma unroll
                        for (int u = 0; u < MW; ++u) Y = hv[u] ? fma(-rj[u], vj[u], Y) : Y;
                    }
                    B[SL::Y + lane] = Y;
                    LSYNC();
                    nu = U;
#pragma unroll
                    for (int kg = 0; kg < 3; ++kg) {   // 10 loads in flight per group
                        double yv[5], lt[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) { yv[u] = B[SL::Y + 3 * (5 * kg + u) + ax_l]; lt[u] = Lt[k_l * 15 + 5 * kg + u]; }
#pragma unroll
                        for (int u = 0; u < 5; ++u) nu = fma(lt[u], yv[u], nu);
                    }
                    LSYNC();   // (slack-free variants: nu overwrites r, Y overwrote x)
                    B[SL::NU + lane] = nu;
                    LSYNC();
                }
                return nu;
            };
            // z = H^-1 nu (za) and Lambda z (zw) for this lane's component, from B[NU]
            auto direction = [&](double &za, double &zw) {
                za = 0.0; zw = 0.0;
                if (comp) {
#pragma unroll
                    for (int kg = 0; kg < 3; ++kg) {
                        double nk[5], th[5], tm[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const int kk = 5 * kg + u;
                            nk[u] = B[SL::NU + 3 * kk + ax_l]; th[u] = G[k_l * 30 + kk]; tm[u] = G[(15 + k_l) * 30 + kk];
                        }
#pragma unroll
                        for (int u = 0; u < 5; ++u) { za = fma(th[u], nk[u], za); zw = fma(tm[u], nk[u], zw); }
                    }
                }
            };
            // x(lambda) re-derived from the multipliers: nu = -N_W lambda, a = a_unc + H^-1 nu, w = w_unc + Lambda H^-1 nu,
            // eps = -(st + sum lambda sigma)/2 for live slack rows
            auto primal_fast = [&]() {
                pristine = false;
                B[SL::RR + lane] = (lane < q) ? B[SL::SLAM + lane] : 0.0;
                unsigned long long smk = 0ull;
                if (soft) {
                    const int mt = (lane < q) ? ((s_meta[lane] >> 8) & 0xff) : -1;
                    smk = __ballot(mt >= TY_COLL && B[SL::SSS + (lane < q ? lane : 0)] != 0.0);
                }
                LSYNC();
                const double nu = residual(0.0, 0.0);
                if (!soft) g_l = nu;   // H x(lambda) + f = -N_W lambda
                double za, zw;
                direction(za, zw);
                if (comp) {
                    a = a_unc + za; w = (SOFT ? w_unc : B[SL::WU + lane]) + zw;
                    B[SL::A + lane] = a; B[SL::W + lane] = w;
                }
                if (soft) {
                    for (int i = lane; i < nr; i += 64) {
                        const int fl = r_fl[i];
                        if (!(fl & RF_LIVE)) r_eps[i] = 0.0;
                        else if (!(fl & (RF_COLL | RF_SLKU | RF_SLKL))) r_eps[i] = -0.5 * r_st[i];
                    }
                    const bool mine = (smk >> lane) & 1ull;
                    const int myrow = mine ? (s_meta[lane] >> 16) : -1;
                    bool owner = false;
                    double acc = 0.0;
                    if (mine) {
                        int s0, s1, s2;
                        row_slots(myrow, s0, s1, s2);
                        acc += B[SL::RR + s0] * B[SL::SSS + s0];
                        if (s1 != 0xff) acc += B[SL::RR + s1] * B[SL::SSS + s1];
                        if (s2 != 0xff) acc += B[SL::RR + s2] * B[SL::SSS + s2];
                        owner = lane == s0;
                    }
                    // (the row's linear cost from the register of the lane that holds the row -- rows 0 .. 63 --, all lanes in the shuffle)
                    double stv = __shfl(rcst[0], (mine && myrow < 64) ? myrow : 0);
                    if (owner) { if (myrow >= 64) stv = r_st[myrow]; r_eps[myrow] = -0.5 * (stv + acc); }
                }
                LSYNC();
            };
            // value n_j'x - d_j of the slot owned by this lane
            auto slot_value = [&](int j) -> double {
                const int meta = s_meta[j], gi = meta & 0xff;
                const int kb = gi >= 15 ? gi - 15 : gi;
                const double *base = B + (gi >= 15 ? SL::W : SL::A) + 3 * kb;
                double v = B[SL::SVEC + 3 * j] * base[0] + B[SL::SVEC + 3 * j + 1] * base[1] + B[SL::SVEC + 3 * j + 2] * base[2];
                if (soft) { const double ss = B[SL::SSS + j]; if (((meta >> 8) & 0xff) >= TY_COLL && ss != 0.0) v += ss * r_eps[meta >> 16]; }
                return v - B[SL::SD + j];
            };
            auto write_slot = [&](const Cd &p, double lam) {
                if (lane == 0) {
                    B[SL::SVEC + 3 * q] = p.v0; B[SL::SVEC + 3 * q + 1] = p.v1; B[SL::SVEC + 3 * q + 2] = p.v2;
                    B[SL::SD + q] = p.d; B[SL::SLAM + q] = lam;
                    if (soft) B[SL::SSS + q] = p.ss;
                    s_meta[q] = p.gi | (p.ty << 8) | (p.idx << 16);
                }
            };
            auto slack_desc = [&](int ty, int idx) -> Cd {
                Cd c;
                c.ty = ty; c.idx = idx; c.gi = 0; c.si = idx; c.v0 = c.v1 = c.v2 = 0.0;
                if (ty == TY_SLKU) { c.ss = 1.0; c.d = 0.0; } else { c.ss = -1.0; c.d = -r_slb[idx]; }
                return c;
            };

            // delete slot l from the working set, with the bookkeeping of its kind; the pin of a soft row whose collision row just
            // left the set is de-instantiated with it (not for row `keep_row`: the entering constraint's own row)
            auto drop_slot = [&](int l, int keep_row) {
                const int dmeta = UNI(s_meta[l]);
                const int dty = (dmeta >> 8) & 0xff, didx = dmeta >> 16;
                LSYNC();
                if (dty >= TY_COLL) {
                    if (!soft && didx < 64 * RC) { if (lane == (didx & 63)) rcfl &= ~(1u << (didx >> 6)); }
                    else if (lane == 0) {
                        const int bit = (dty == TY_COLL) ? RF_COLL : (dty == TY_SLKU ? RF_SLKU : RF_SLKL);
                        if (soft) r_fl[didx] &= ~bit; else r_bits[didx >> 5] &= ~(1u << (didx & 31));
                    }
                }
                remove_slot2<SOFT, QCAP, PERSIST, TS, TF>(B, lane, q, l, cslot, cm, xo, m_row, nrmax);
                if (soft && dty == TY_COLL && didx != keep_row) {
                    const int fl = UNI((int)r_fl[didx]);
                    if ((fl & RF_LIVE) && (fl & RF_SLKU) && !(fl & RF_SLKL)) {
                        const int mm = (lane < q) ? s_meta[lane] : 0;
                        const unsigned long long um = __ballot(lane < q && ((mm >> 8) & 0xff) == TY_SLKU && (mm >> 16) == didx);
                        const int ul = __ffsll((long long)um) - 1;
                        LSYNC();
                        if (lane == 0) { r_fl[didx] = 0; r_eps[didx] = 0.0; }
                        remove_slot2<SOFT, QCAP, PERSIST, TS, TF>(B, lane, q, ul, cslot, cm, xo, m_row, nrmax);
                        nlive--;
                    }
                }
            };

            // Agents with more rows on their step than the in-loop certificate can look at (38): the level is certified BEFORE the solve, while
            // the factor's block is free to hold the planes of all of them -- when it passes, the search ends at the first line of the first batch
            // that meets the polytope (microseconds); when it fails, a whole solve that would have ended in the dual method's own proof is saved.
            bool pre_inf = false;
            if (SOFT && ladder && violation && !cert_known && !hp && nr > 38 && q == 0) {
                constexpr int CP = (t_doubles(TS) * (int)sizeof(TF) / 8) / 4 < 134 ? (t_doubles(TS) * (int)sizeof(TF) / 8) / 4 : 134;
                cost += 176;
                pre_inf = uni_b(ladder_level_infeasible(r_xi, r_b, r_sd, r_slb, r_kc, nr, B + SL::T, P.h, P.alim, 1.0, whi_l, wlo_l, lane, CP));
                LSYNC();
                for (int i = lane; i < 4 * ((nr < CP - 6 ? nr : CP - 6) + 6); i += 64) B[SL::T + i] = 0.0;
                LSYNC();
                cert_known = true;   // (this level is not tested again)
                if (pre_inf) rc = 1;
            }
            // Crash start, first batch: every acceleration bound violated at the unconstrained minimiser is appended BEFORE the iteration
            // starts (crash_append: from the table while the sets are prefixes of the horizon, else one bound per axis and call).  Out of
            // line and outside the loop: inside it the call cost the iteration 3-4 % (registers live across the call site).  Bounds that
            // only show after the multipliers of this batch are solved go through the one-at-a-time appends of the iteration.
            if (SOFT && crash && !pre_inf) {
                bool tbl_ok = true;
                while (q < (QCAP - 4 < 44 ? QCAP - 4 : 44)) {
                    // (the function's own first test, here: the call that only finds nothing left costs an out-of-line call with its register traffic)
                    if (__ballot(comp && !(cslot & 0x30000u) && fabs(a) - P.alim > tol) == 0ull) break;
                    const CrashRes cr = crash_append<SOFT, QCAP, PERSIST, TF, TS>((LdsD *)B, (const LdsD *)G, P.tables + TAB_DOUBLES + (size_t)ccase * 2 * TAB_TP_CASE,
                                                                          lane, q, cslot, a, P.alim, tol, tbl_ok);
                    const int crf = UNI(cr.flags);   // (an out-of-line function returns in vector registers: the slot count would be lane-dependent from here on)
                    const int nb = crf >> 8;
                    if (crf & 2) crash_stop = true;
                    if (nb == 0) break;
                    iters += nb; nfast += nb;
#ifdef DMPC_DEV_TRACE
                    if (crf & 4) dev_tbl += nb; else dev_gen += nb;
#endif
                    q = UNI(cr.q); cslot = cr.cslot;
                    tbl_ok = (crf & 1) != 0;
                    if (crash_stop) break;
                }
                if (q > maxq) maxq = q;
                LSYNC();
                PH(17);   // (development: the crash start's first batch on its own)
            }
            bool fresh = !warm && nfast == 0;   // primal == x(lambda) with refined lambda
            int since_sync = 0;
            // Round 4 (found by the randomized campaign: one solveSoftDMPCall agent-step of 135 000 ended one ladder level late, on a level an
            // LP shows feasible with room to spare).  An acceleration bound violated by 1.4e-9 -- drift of the incrementally updated iterate,
            // the bound is exactly active at the vertex -- was picked as pivot, found dependent on the working set, and the ratio test then
            // took entries of r that are pure round-off of the factor (1e-19) as blocking constraints: partial steps of 1e19, the
            // multipliers destroyed, "infeasible".  Two guards: on a DEPENDENT pivot only entries of r above 1e-9 of its largest can block,
            // and an infeasibility verdict reached without a step on an iterate that has not been re-derived from the multipliers since
            // the last primal step is not believed: the iterate is re-derived and refined (`resync`), and the violations are looked at again.
            bool x_synced = fresh, resync = false;
            int resyncs = 0;
            bool cert_done = cert_known;   // (a level the ladder step below already put through the certificate is not tested again: 44 us a call for 26 rows)
            for (; !pre_inf;) {
                if (soft && ladder && violation && !cert_done && iters - nfast >= LADDER_CERT_AFTER) {
                    cert_done = true;
                    PH(4); PHC(12);
                    cost += 176;   // (44 us a call)
                    static_assert(!SOFT || (SL::XS == SL::RR + 64 && SL::NU == SL::XS + 64), "the three staging vectors are one block of 176 doubles");
                    const bool cert_inf = uni_b(ladder_level_infeasible(r_xi, r_b, r_sd, r_slb, r_kc, nr, B + SL::RR, P.h, P.alim, 1.0, whi_l, wlo_l, lane, 44));   // (RR | XS | NU are dead between two iterations: 38 rows of a step)
                    PH(7);
                    if (cert_inf) { rc = 1; break; }
                }
                // ---- most violated constraint not in the working set (score = violation / |n|_{H^-1}, fp32)
                double bestv = 0.0; float bests = 0.f; int bestc = -1;
#define CAND(v_, w_, code_) do { const double v__ = (v_); const float s__ = (float)v__ * (w_); \
                                 if (v__ > tol && s__ > bests) { bests = s__; bestv = v__; bestc = (code_); } } while (0)
                // the crash ends its batch when nothing is left to append, when a multiplier had to be dropped, or short of the capacity
                const bool crash_finish = crash && (crash_stop || q >= (QCAP - 4 < 44 ? QCAP - 4 : 44));   // (the same for the 48- and 64-slot kernels: the path of an agent must not depend on the tier)
                if (comp && !crash_finish) {
                    // a <= alim and -a <= alim are violated one at a time: one candidate for the pair (not while a member is active)
                    const bool hi = a > 0.0;
                    if (!(cslot & 0x30000u)) CAND(fabs(a) - P.alim, wbox_f, ((hi ? TY_BOXHI : TY_BOXLO) << 16) | lane);
                    const double c2 = w - whi_l, c3 = wlo_l - w;
                    if (!crash && !(cslot & 0xc0000u)) CAND(fmax(c2, c3), wpos_f, ((c2 > c3 ? TY_POSHI : TY_POSLO) << 16) | lane);
                }
                if (!crash) {
#pragma unroll
                for (int c = 0; c < RC; ++c) {   // rows held in registers
                    const int i = lane + 64 * c;
                    if (i < nr) {
                        const int fl = soft ? (int)r_fl[i] : (int)((rcfl >> c) & 1u), kc = rckc[c];
                        double v = -(rcx0[c] * B[SL::W + 3 * kc] + rcx1[c] * B[SL::W + 3 * kc + 1] + rcx2[c] * B[SL::W + 3 * kc + 2]) - rcb[c];
                        if (soft && hp) v += rcsd[c] * rcslb[c];   // (level check: every slack at its bound)
                        if (soft && (fl & RF_LIVE)) {
                            const double e = r_eps[i];
                            v += rcsd[c] * e;
                            if (!(fl & RF_SLKU)) CAND(e, wslk_u, (TY_SLKU << 16) | i);
                            const double lo = rcslb[c] - e;   // -eps <= -slb
                            if (!(fl & RF_SLKL)) CAND(lo, wslk_l, (TY_SLKL << 16) | i);
                        }
                        if (!(fl & RF_COLL)) CAND(v, rcw[c], (TY_COLL << 16) | i);
                    }
                }
                for (int i = lane + 64 * RC; i < nr; i += 64) {   // the rest streams from the global scratch
                    const int fl = soft ? (int)r_fl[i] : (int)((r_bits[i >> 5] >> (i & 31)) & 1u), kc = r_kc[i];
                    const double x0 = r_xi[3 * i], x1 = r_xi[3 * i + 1], x2 = r_xi[3 * i + 2];
                    double v = -(x0 * B[SL::W + 3 * kc] + x1 * B[SL::W + 3 * kc + 1] + x2 * B[SL::W + 3 * kc + 2]) - r_b[i];
                    if (soft && hp) v += r_sd[i] * r_slb[i];
                    if (soft && (fl & RF_LIVE)) {
                        const double e = r_eps[i];
                        v += r_sd[i] * e;
                        if (!(fl & RF_SLKU)) CAND(e, wslk_u, (TY_SLKU << 16) | i);
                        const double lo = r_slb[i] - e;
                        if (!(fl & RF_SLKL)) CAND(lo, wslk_l, (TY_SLKL << 16) | i);
                    }
                    if (!(fl & RF_COLL) && v > tol) CAND(v, row_weight(x0, x1, x2, kc, soft ? r_sd[i] : 0.0), (TY_COLL << 16) | i);
                }
                }
#undef CAND
                const float smax = wave_max_f(bests);
                const bool forced = SOFT && resync;   // (re-derive the iterate first: the scan above ran on the stale one)
                resync = false;
                const unsigned long long wm = forced ? 0ull : __ballot(bestc >= 0 && bests == smax);
                PH(0);
                if (wm == 0ull) {
                    PHC(10);
                    // (not inside the crash start: its scans look at the acceleration bounds only, and the first of them -- right after the table batch, at
                    // the very BEGINNING of a solve -- finds nothing: until round 5 every agent with a crash start claimed its wave's next queue position
                    // there, and that position then waited behind the whole solve: the queue degenerated to an assignment one agent ahead, measured
                    // launch 880 us where list scheduling of the same durations in the same order gives 770)
                    if (!forced && !crash) CLAIM_NEXT();
                    if (!crash && !forced && (q == 0 || fresh)) break;   // optimal
                    if (!crash && !soft && !F32T && !forced) {   // (fp32 factor: the incrementally updated iterate drifts by ~1e-7 per step -- always the full verification)
                        // (slack-free variants; the slack variants carry multipliers of 1e5-1e6 and always take the full verification)
                        // No constraint is violated at the (incrementally updated) iterate.  Round-off of the factor reaches the iterate
                        // only through N_W: an error dr of r = T T's moves x by t H^-1 N_W dr, which shows in the values of the
                        // active constraints (N_W' H^-1 N_W is non-singular).  If those are zero to round-off the iterate IS the
                        // minimiser over its working set and nothing has to be re-derived (one reduction instead of the primal from the
                        // multipliers, the refinement and a second violation scan)
                        const double rho0 = (lane < q) ? slot_value(lane) : 0.0;
                        if (!(wave_max0(fabs(rho0)) > 1e-13)) break;
                    }
                    const bool was_fresh = fresh;
                    // verification: primal from the multipliers, refine the active-set residual, re-check
                    // (crash: the appended slots carry lambda = 0 and their violation at the stale iterate -- the same pass solves them)
                    if (!fresh) {
                    // (pristine: nothing but appends without a step since the try began -- every multiplier is zero and the iterate in registers and LDS
                    // IS x(lambda) = the unconstrained minimiser: re-deriving it would write the same bits)
                    if (!(SOFT && pristine)) primal_fast();
                    double mx_prev = INFINITY;
                    for (int pass = 0; pass < REFINE_PASSES; ++pass) {
                        const double rho = (lane < q) ? slot_value(lane) : 0.0;
                        const double mx = wave_max0(fabs(rho));
                        if (!(mx > 1e-13)) break;
                        if (F32T && !(mx < 0.5 * mx_prev)) break;   // (fp32 factor: the refinement has stopped contracting)
                        mx_prev = mx;
                        B[SL::XS + lane] = rho; LSYNC();
                        const double dvj = t_tmul2<QCAP, SL::T, SL::XS, TS, TF, SOFT>(B, lane, q, xo);
                        B[SL::RR + lane] = dvj; LSYNC();
                        const double ri = t_mul2<QCAP, SL::T, SL::RR, TS, TF, SOFT>(B, lane, q, xo);
                        if (lane < q) B[SL::SLAM + lane] += ri;
                        LSYNC();
                        primal_fast();
                    }
                    }
                    if ((SOFT || F32T) && !crash && !was_fresh && accept_drops < 4) {   // (slack variants, and every variant with the fp32 factor; the fp64 slack-free kernels keep `crash` a compile-time false)
                        // acceptance: the refined multipliers must be non-negative.  At a degenerate vertex the incrementally updated
                        // multipliers can drift and the refinement then uncovers a negative one: the point is feasible but not the
                        // minimiser (randomized campaign, seed 2: one solveSoftDMPCall agent-step of 454 611, objective off by 2e-5).
                        // Such a constraint is dropped through the finish of the crash start and the iteration goes on.
                        const double lam = lane < q ? B[SL::SLAM + lane] : 0.0;
                        const double lmax = wave_max0(fabs(lam));
                        // (round 5: the noise floor of the refined multipliers is 1e-9 absolute + 1e-11 of the largest one; it was 1e-9 (1 + largest).  With
                        // penalties of 1.6e7 -- solveSoftDMPCall on retry-ladder level 5 in the C5 box -- the old floor hid multipliers down to
                        // -0.016: campaign seed 73, scene 756, one agent of 3.9 M kept a constraint with a multiplier of about -5e-3 and ended 4.8e-4 m
                        // off the minimiser (objective higher by 7e-6, stationarity 1.3e-9 instead of 1e-15; tests/dev/gpu_campaign_scene.py 73 756 all3;
                        // the round-4 library gives the same wrong answer).)
#ifndef DMPC_ACC_REL
#define DMPC_ACC_REL 1e-11
#endif
                        if (__any(lane < q && lam < -(1e-9 + DMPC_ACC_REL * lmax))) { crash = true; crash_stop = true; crash_box = false; accept_drops++; }
                    }
                    if (crash) {
                        unsigned long long neg = __ballot(lane < q && B[SL::SLAM + (lane < q ? lane : 0)] < 0.0);
                        if (neg != 0ull) {   // guessed bounds that do not belong: drop them (highest slot first), solve again, then leave the crash
                            LSYNC();
                            if (crash_box) {
                                while (neg != 0ull) {
                                    const int l = 63 - __clzll((long long)neg);
                                    neg &= ~(1ull << l);
                                    remove_slot2<SOFT, QCAP, PERSIST, TS, TF>(B, lane, q, l, cslot, cm, xo, m_row, nrmax);
                                }
                            } else drop_slot(63 - __clzll((long long)neg), -1);   // any kind of slot (and maybe its pin): one per pass
#ifdef DMPC_DEV_TRACE
                            dev_negdrops++;
#endif
                            crash_stop = true; fresh = false;
                            PH(6);
                            continue;
                        }
                        if (was_fresh || crash_stop || ++crash_rounds >= 8) {
                            crash = false;
                            if (!soft) dual = dual0 + 0.5 * wave_sum0(g_l * (a - a_unc));   // cost at x(lambda): f(x_unc) + 1/2 nu' H^-1 nu
                        }
                    }
                    fresh = true; x_synced = true;
                    PH(6);
                    continue;
                }
                fresh = false;
                const int src = __ffsll((long long)wm) - 1;
                const int pcode = readlane_i(bestc, src);
                double vp = readlane_d(bestv, src);
                Cd p;
                {
                    const int pty = pcode >> 16, pidx = pcode & 0xffff;
                    p.ty = pty; p.idx = pidx; p.gi = 0; p.si = -1; p.v0 = p.v1 = p.v2 = 0.0; p.ss = 0.0; p.d = 0.0;
                    if (pty < TY_COLL) {
                        const int k = pidx / 3, ax = pidx - 3 * k;
                        const double sgn = (pty == TY_BOXHI || pty == TY_POSHI) ? 1.0 : -1.0;
                        p.gi = (pty < TY_POSHI) ? k : 15 + k;
                        p.v0 = ax == 0 ? sgn : 0.0; p.v1 = ax == 1 ? sgn : 0.0; p.v2 = ax == 2 ? sgn : 0.0;
                        p.d = (pty < TY_POSHI) ? P.alim : ((pty == TY_POSHI) ? readlane_d(whi_l, pidx) : -readlane_d(wlo_l, pidx));
                    } else if (pidx < 64 * RC) {   // row data from the owning lane's registers
                        const int ol = pidx & 63;
                        const bool hi = RC > 1 && pidx >= 64;
                        if (pty == TY_COLL) {
                            p.gi = 15 + readlane_i(hi ? rckc[1] : rckc[0], ol);
                            p.v0 = -readlane_d(hi ? rcx0[1] : rcx0[0], ol); p.v1 = -readlane_d(hi ? rcx1[1] : rcx1[0], ol);
                            p.v2 = -readlane_d(hi ? rcx2[1] : rcx2[0], ol); p.d = readlane_d(hi ? rcb[1] : rcb[0], ol);
                            if (soft && !hp) { p.si = pidx; p.ss = readlane_d(hi ? rcsd[1] : rcsd[0], ol); }
                            if (soft && hp) p.d -= readlane_d(hi ? rcsd[1] : rcsd[0], ol) * readlane_d(hi ? rcslb[1] : rcslb[0], ol);
                        } else if (pty == TY_SLKU) { p.si = pidx; p.ss = 1.0; }
                        else { p.si = pidx; p.ss = -1.0; p.d = -readlane_d(hi ? rcslb[1] : rcslb[0], ol); }
                    } else if (pty == TY_COLL) {
                        p.gi = 15 + r_kc[pidx];
                        p.v0 = -r_xi[3 * pidx]; p.v1 = -r_xi[3 * pidx + 1]; p.v2 = -r_xi[3 * pidx + 2];
                        p.d = r_b[pidx];
                        if (soft && !hp) { p.si = pidx; p.ss = r_sd[pidx]; }
                        if (soft && hp) p.d -= r_sd[pidx] * r_slb[pidx];
                    } else p = slack_desc(pty, pidx);
                }
                // lazily instantiate the eps<=0 pin of a soft row that becomes active (S(u,u) = 1/2)
                if (soft && !hp && p.ty == TY_COLL && !(UNI((int)r_fl[p.idx]) & RF_LIVE)) {
                    if (q >= QCAP - 1) { rc = 2; break; }
                    const Cd u = slack_desc(TY_SLKU, p.idx);
                    ENSURE_EXT(q);
                    if (lane < ((q + 8) & ~7)) ((TF *)(B + SL::T))[tcol(q) + ((TS < QCAP && q >= TS) ? xo : 0) + lane] = (TF)((lane == q) ? 1.4142135623730951 : 0.0);
                    if (lane == 0) { r_fl[p.idx] |= (RF_LIVE | RF_SLKU); m_row[nrmax + p.idx] = (unsigned char)q; }
                    nlive++;
                    write_slot(u, p.idx < 64 ? -readlane_d(rcst[0], p.idx) : -r_st[p.idx]);
                    q++;
                    LSYNC();
                }
                PH(1);
                double lam_p = 0.0;
                // n_p' H^-1 n_p
                const double spp = G[p.gi * 31] * (p.v0 * p.v0 + p.v1 * p.v1 + p.v2 * p.v2) + ((soft && p.si >= 0) ? 0.5 * p.ss * p.ss : 0.0);
                const bool p_isA = p.ty < TY_POSHI;
                // ---- inner loop: partial steps until p can be added
                for (;;) {
                    if (++iters > P.iter_cap) { rc = 3; break; }
                    cost += crash ? 3 + (q >> 3) : 8 + (q >> 2);   // (a lone wave: 2 us per iteration + 0.06 us per slot; an append without a step: a third)
                    // s = N_W' H^-1 n_p on the slot lanes
                    double sv = 0.0;
                    int mymeta = 0;
                    if (lane < q) {
                        mymeta = s_meta[lane];
                        const int gj = mymeta & 0xff;
                        const double dot3 = B[SL::SVEC + 3 * lane] * p.v0 + B[SL::SVEC + 3 * lane + 1] * p.v1 + B[SL::SVEC + 3 * lane + 2] * p.v2;
                        sv = G[gj * 30 + p.gi] * dot3;
                        if (soft) {
                            const double ssj = B[SL::SSS + lane];
                            if (((mymeta >> 8) & 0xff) >= TY_COLL && p.si >= 0 && (mymeta >> 16) == p.si && ssj != 0.0) sv += 0.5 * ssj * p.ss;
                        }
                    }
                    B[SL::XS + lane] = sv; LSYNC();
                    PH(13);
                    const double dvj = t_tmul2<QCAP, SL::T, SL::XS, TS, TF, SOFT>(B, lane, q, xo);
                    B[SL::RR + lane] = dvj; LSYNC();
                    PH(14);
                    const double ri = t_mul2<QCAP, SL::T, SL::RR, TS, TF, SOFT>(B, lane, q, xo);
                    LSYNC();
                    PH(2);
                    if (crash) {   // append without a step: column [-r/rho; 1/rho] with rho^2 = s_pp - |T's|^2, lambda_p = 0 until the batch is solved
                        const double dlt = spp - wave_sum0(dvj * dvj);
                        if (!(dlt > 1e-9 * spp)) { crash_stop = true; break; }   // (distinct bounds are independent; guard only)
                        const double irho = fast_rsq(dlt);
                        if (lane < ((q + 8) & ~7)) ((TF *)(B + SL::T))[tcol(q) + lane] = (TF)((lane < q) ? (-ri * irho) : ((lane == q) ? irho : 0.0));
                        write_slot(p, 0.0);
                        if (lane == p.idx) cslot = (cslot & ~0xffu) | (unsigned)q | (p.ty == TY_BOXHI ? 0x10000u : 0x20000u);
                        q++; nfast++;
                        if (q > maxq) maxq = q;
                        LSYNC();
                        break;
                    }
                    B[SL::RR + lane] = ri;
                    unsigned long long smk = 0ull;
                    if (soft) smk = __ballot(lane < q && ((mymeta >> 8) & 0xff) >= TY_COLL && B[SL::SSS + (lane < q ? lane : 0)] != 0.0);
                    LSYNC();
                    // residual nu = n_p - N_W r (explicit: when p is nearly dependent on W, nu is small and the round-off of r
                    // enters delta squared -- this makes the dependence / infeasibility test reliable), z = H^-1 nu, delta = nu'z
                    double pU = 0.0, pY = 0.0;
                    if (comp && p.ty <= TY_COLL && (p.gi >= 15 ? p.gi - 15 : p.gi) == k_l) {
                        const double vpax = ax_l == 0 ? p.v0 : (ax_l == 1 ? p.v1 : p.v2);
                        if (p_isA) pU = vpax; else pY = vpax;
                    }
                    const double nu = residual(pU, pY);
                    double za, zw;
                    direction(za, zw);
                    double part = nu * za;
                    // slack part: nu_eps(row) = sigma_p[si_p==row] - sum_j r_j sigma_j[si_j==row]; H_eps^-1 = 1/2
                    double nue = 0.0; bool owner = false;
                    bool p_row_has_slot = false;
                    if (soft) {
                        const bool mine = (smk >> lane) & 1ull;
                        const int myrow = mine ? (mymeta >> 16) : -1;
                        if (mine) {
                            int s0, s1, s2;
                            row_slots(myrow, s0, s1, s2);
                            if (p.si == myrow) nue += p.ss;
                            nue -= B[SL::RR + s0] * B[SL::SSS + s0];
                            if (s1 != 0xff) nue -= B[SL::RR + s1] * B[SL::SSS + s1];
                            if (s2 != 0xff) nue -= B[SL::RR + s2] * B[SL::SSS + s2];
                            owner = lane == s0;
                        }
                        if (p.si >= 0) p_row_has_slot = (UNI((int)(m_row[p.si] & m_row[nrmax + p.si] & m_row[2 * nrmax + p.si])) & 0xff) != 0xff;
                        if (owner) part += 0.5 * nue * nue;
                        if (p.si >= 0 && !p_row_has_slot && lane == 63) part += 0.5 * p.ss * p.ss;
                    }
                    const double delta = wave_sum0(part);
                    PH(3);
                    // more active constraints than variables is impossible: whatever round-off says, a constraint picked when
                    // the working set already spans all 45 + nlive variables is dependent
                    const bool dependent = !(delta > DEP_TOL * spp) || q >= N3 + nlive;
                    const double t2 = dependent ? INFINITY : fast_div(vp, delta);
                    // ratio test on the multipliers: the blocking slot maximises r_j / lambda_j (identity 0: one bound_ctrl
                    // DPP maximum instead of a minimum with an infinity identity); t1 = 1 / max
                    const double lam_l = B[SL::SLAM + lane];
                    // (slack variants only: the slack-free kernels sit at their register limit -- both guards cost the headline 8 % -- and four rounds of
                    // campaigns have not shown the failure there: their rows are not nearly parallel triples)
                    const double rcut = (SOFT && dependent) ? 1e-9 * wave_max0((lane < q) ? fabs(ri) : 0.0) : 0.0;   // (dependent pivot: round-off of the factor does not block)
                    const double iratio = (lane < q && ri > rcut) ? (lam_l > 1e-300 ? fast_div(ri, lam_l) : INFINITY) : 0.0;
                    const double imax = wave_max0(iratio);
                    const double t1 = imax > 0.0 ? (imax < INFINITY ? fast_rcp(imax) : 0.0) : INFINITY;   // inf when no multiplier decreases
                    const double t = fmin(t1, t2);
                    PH(15);
#ifdef DMPC_DEV_TRACE
                    if (P.dbg && gid == P.dbg_agent && lane == 0 && iters <= P.dbg_cap) {
                        double *d = P.dbg + (size_t)(iters - 1) * 8;
                        d[0] = (double)pcode; d[1] = (double)q; d[2] = delta; d[3] = spp; d[4] = t1; d[5] = t2; d[6] = vp; d[7] = lam_p;
                    }
#endif
                    if (!(t < INFINITY)) {
                        if (SOFT && lam_p == 0.0 && !x_synced && resyncs < 3) { ++resyncs; resync = true; break; }   // not believed on a stale iterate
                        // How far up the retry ladder does THIS proof reach (round 5)?  The verdict is a Farkas combination: the pivot depends on the
                        // working set, n_p = N_W r with no r_j > 0, so y = (1, -r) >= 0 combines the constraints to 0 <= d_p - sum r_j d_j, and that
                        // number is negative.  A ladder step changes nothing but right-hand sides -- the slack bounds -slb 2^t (in the level check of
                        // solveSoftDMPCall: the rows' b - d slb 2^t) -- so the same combination reads C + 2^m U at level t + m: every level where it
                        // is still clearly negative is infeasible by the same proof and is skipped (counted as a try, as the reference would have
                        // spent it).  The longest agents of the solveSoftDMPCall replay proved four levels in a row infeasible with 80 iterations each.
                        if (SOFT && ladder && violation) {
                            double c_l = 0.0, u_l = 0.0;
                            if (lane < q) {
                                const int mt = (mymeta >> 8) & 0xff, mi = mymeta >> 16;
                                const double dj = B[SL::SD + lane];
                                double uj = 0.0;
                                if (hp) { if (mt == TY_COLL) uj = -r_sd[mi] * r_slb[mi]; }
                                else if (mt == TY_SLKL) uj = dj;
                                c_l = -ri * (dj - uj); u_l = -ri * uj;
                            }
                            double up = 0.0;
                            if (hp) { if (p.ty == TY_COLL) up = -r_sd[p.idx] * r_slb[p.idx]; }
                            else if (p.ty == TY_SLKL) up = p.d;
                            const double Cc = (p.d - up) + wave_sum0(c_l), Uc = up + wave_sum0(u_l);
                            lev_skip = 0;
                            if (Cc + Uc < 0.0 && !P.no_level_skip) {   // (the proof as it stands; anything else is round-off: no skipping)
                                double kk = 2.0;
                                while (lev_skip < 40 && Cc + kk * Uc < -1e-7 * (fabs(Cc) + kk * fabs(Uc))) { ++lev_skip; kk *= 2.0; }
                            }
                        }
                        rc = 1; break;
                    }
                    if (lane < q) B[SL::SLAM + lane] -= t * ri;
                    lam_p += t; pristine = false;
                    if (!dependent) {
                        if (!soft) {
                            dual += t * delta * (lam_p - 0.5 * t);
                            if (dual > fbound) { rc = 1; break; }   // no point of the acceleration box costs this much
                            // gradient of the cost at the iterate, g = H x + f = -(N_W lambda + n_p lambda_p) with lambda >= 0:
                            // every feasible point a satisfies g.a >= g.x; when even the best point of the box misses that,
                            // alim |g|_1 < g.x, the multipliers are a Farkas certificate of infeasibility
                            g_l -= t * nu;
                            if (iters >= FARKAS_AFTER) {
                                const double an = a - t * za;
                                const double ga = g_l * an, gb = P.alim * fabs(g_l);
                                if (wave_sum0(ga - gb - 1e-6 * (fabs(ga) + gb)) > 0.0) { rc = 1; break; }
                            }
                        }
                        vp -= t * delta;
                        a -= t * za; w -= t * zw;
                        if (SOFT) x_synced = false;
                        if (comp) { B[SL::A + lane] = a; B[SL::W + lane] = w; }
                        if (soft) {
                            if (owner) r_eps[mymeta >> 16] -= t * 0.5 * nue;
                            if (p.si >= 0 && !p_row_has_slot && lane == 63) r_eps[p.si] -= t * 0.5 * p.ss;
                        }
                    }
                    PH(16);
                    if (t2 <= t1) {
                        // full step: append p (new column of T = [-r/rho ; 1/rho], zero below the diagonal)
                        if (q >= QCAP) { rc = 2; break; }
                        const double irho = fast_rsq(delta);
                        ENSURE_EXT(q);
                        if (lane < ((q + 8) & ~7)) ((TF *)(B + SL::T))[tcol(q) + ((TS < QCAP && q >= TS) ? xo : 0) + lane] = (TF)((lane < q) ? (-ri * irho) : ((lane == q) ? irho : 0.0));
                        write_slot(p, lam_p);
                        if (p.ty < TY_COLL) {
                            if (lane == p.idx) {
                                if (p_isA) cslot = (cslot & ~0xffu) | (unsigned)q | (p.ty == TY_BOXHI ? 0x10000u : 0x20000u);
                                else cslot = (cslot & ~0xff00u) | ((unsigned)q << 8) | (p.ty == TY_POSHI ? 0x40000u : 0x80000u);
                            }
                        } else {
                            if (!soft && p.idx < 64 * RC) { if (lane == (p.idx & 63)) rcfl |= 1u << (p.idx >> 6); }
                            else if (lane == 0) {
                                const int bit = (p.ty == TY_COLL) ? RF_COLL : (p.ty == TY_SLKU ? RF_SLKU : RF_SLKL);
                                if (soft) { r_fl[p.idx] |= bit; if (p.ss != 0.0) m_row[(size_t)(p.ty - TY_COLL) * nrmax + p.idx] = (unsigned char)q; }
                                else r_bits[p.idx >> 5] |= 1u << (p.idx & 31);
                            }
                            if (p.ty == TY_COLL && comp && k_l == p.gi - 15) cm |= 1ull << q;
                        }
                        q++;
                        if (q > maxq) maxq = q;
                        LSYNC();
                        PH(4);
                        break;
                    }
                    PH(4); PHC(11);
                    // partial step: drop the blocking constraint
                    const unsigned long long bm = __ballot(lane < q && ri > rcut && iratio == imax);
                    const int l = __ffsll((long long)bm) - 1;
                    drop_slot(l, (p.ty == TY_COLL) ? p.idx : -1);
                    LSYNC();
                    PH(5);
                }
                if (rc) break;
                if (SOFT && resync) continue;
                // pin added while its collision row is not active: decoupled again -> drop both
                if (soft && p.ty == TY_SLKU && !(UNI((int)r_fl[p.idx]) & (RF_COLL | RF_SLKL))) {
                    const int mm = (lane < q) ? s_meta[lane] : 0;
                    const unsigned long long um = __ballot(lane < q && ((mm >> 8) & 0xff) == TY_SLKU && (mm >> 16) == p.idx);
                    const int ul = __ffsll((long long)um) - 1;
                    LSYNC();
                    if (lane == 0) { r_fl[p.idx] = 0; r_eps[p.idx] = 0.0; }
                    remove_slot2<SOFT, QCAP, PERSIST, TS, TF>(B, lane, q, ul, cslot, cm, xo, m_row, nrmax);
                    nlive--;
                }
                if (((++since_sync) & 31) == 0) primal_fast();   // periodic re-sync with x(lambda)
                PH(4);
            }
            iters_total += iters;
#ifdef DMPC_DEV_TRACE
            dev_nfast += nfast; dev_rounds += crash_rounds;
            if (ph_on && lane == 0 && P.dbg_cap >= 16 && dev_try < 8) {   // development: one record per solve of the traced agent (rows cap-12 .. cap-5 of the trace)
                double *d = P.dbg + (size_t)(P.dbg_cap - 12 + dev_try) * 8;
                d[0] = (double)tries; d[1] = (double)iters; d[2] = (double)nfast; d[3] = (double)rc; d[4] = (double)q; d[5] = hp ? 1.0 : 0.0; d[6] = (double)scale_pow; d[7] = 0.0;
            }
            ++dev_try;
#endif
            if (hp) {
                // level check over: feasible (or undecided: out of slots / iterations) -> the real solve of this level; infeasible -> the
                // ladder step below, as after a failed try
                if (rc != 1) { level_checked = true; if (rc == 0) cert_known = true; tries--; continue; }
            }
            if (rc == 0) { solved = true; break; }
            if (SOFT && rc == 2 && P.qover_bit == ST_CAPACITY && !alt_rule) {   // last tier out of slots: the level once more, rows as sparingly as bounds
                alt_rule = true;
                if (comp) { wbox_f = __builtin_amdgcn_rsqf((float)G[k_l * 31]); wpos_f = __builtin_amdgcn_rsqf((float)G[(15 + k_l) * 31]); }
#pragma unroll
                for (int c = 0; c < RC; ++c) rcw[c] = row_weight(rcx0[c], rcx1[c], rcx2[c], rckc[c], rcsd[c]);
                tries--;
                continue;
            }
            if (rc == 2) {   // first tier: flag for the second-tier launch, rows back to what the scan wrote (exact: powers of two); last tier: capacity error
                status |= P.qover_bit;
                if (soft && scale_pow && P.qover_bit == ST_QOVER) {
                    const double f = ldexp(1.0, -scale_pow);
                    for (int i = lane; i < nr; i += 64) { r_slb[i] *= f; r_st[i] *= f; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
                // tier 1 appends the agent to the list the tier-2 launch works through (order irrelevant: scheduling only)
                if (lane == 0 && P.flag_list && P.qover_bit == ST_QOVER) P.flag_list[atomicAdd(P.flag_count, 1)] = gid;
                break;
            }
            if (rc == 3) { status |= ST_ITERCAP; break; }
            // infeasible: retry ladder (solveSoftDMPCbound.m:147-153): lb_eps *= 2, term *= 2
            if (soft && ladder && violation) {
                PH(4);
                double f = 2.0;
                cert_known = false;
                // (between two solves the inverse factor is dead: its block holds the planes of up to 128 rows; zeroed again afterwards -- every value
                // the products can read stays finite whatever the factor's storage type makes of these bits)
                constexpr int CERT_PLANES = (t_doubles(TS) * (int)sizeof(TF) / 8) / 4 < 134 ? (t_doubles(TS) * (int)sizeof(TF) / 8) / 4 : 134;
                const int cert_used = 4 * ((nr < CERT_PLANES - 6 ? nr : CERT_PLANES - 6) + 6);
                while (tries < max_tries - 1) {
                    if (lev_skip > 0) { --lev_skip; f *= 2.0; ++tries; continue; }   // (infeasible by the proof the failed solve ended with)
                    if (!uni_b(ladder_level_infeasible(r_xi, r_b, r_sd, r_slb, r_kc, nr, B + SL::T, P.h, P.alim, f, whi_l, wlo_l, lane, CERT_PLANES))) { cert_known = true; break; }
                    f *= 2.0; ++tries;
                }
                lev_skip = 0;
                LSYNC();
                for (int i = lane; i < cert_used; i += 64) B[SL::T + i] = 0.0;
                for (int i = lane; i < nr; i += 64) { r_slb[i] *= f; r_st[i] *= f; }
                rcslb[0] *= f; rcslb[1] *= f; rcst[0] *= f; rcst[1] *= f;
                scale_pow += ilogb(f);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                LSYNC();
                PH(7);
                level_checked = false;
                continue;
            }
            if (ladder || var == VAR_REPAIR) tries = (var == VAR_REPAIR && P.max_tries <= 0) ? 10 : max_tries;
            break;
        }
        if (!solved && !(status & (ST_CAPACITY | ST_ITERCAP | ST_QOVER))) status |= ST_INFEAS;
    }

    // ---------------------------------------------------------------- a9/a10: propagate, outputs
    // The launch parameters the output stage needs are read again from the kernel-argument segment (scalar loads through an
    // opaque pointer) instead of staying live in SGPRs across the solver loop, where they were spilled to VGPR lanes.
    const KargPtr Qp = kernarg_params();
    CLAIM_NEXT();
    if (TS < QCAP && xo != 0) {   // the extension goes back to the pool, zeroed (every LDS value a wave can read stays finite)
        double *ex = B + xo + tcol(TS);
        for (int i = lane; i < ext_doubles(QCAP, TS); i += 64) ex[i] = 0.0;
        LSYNC();
        if (lane == 0) atomicOr((unsigned *)(shtab) + (PERSIST_TABLE_BYTES - 16) / 4, 1u << ((xo + tcol(TS) - ext0) / ext_doubles(QCAP, TS)));
    }
    int nslack = 0;
    if (solved) {
        status |= ST_SOLVED;
        if (soft) {
            int cnt = 0;
            for (int i = lane; i < nr; i += 64) cnt += (r_eps[i] < -1e-12) ? 1 : 0;
            nslack = (int)wave_sum0((double)cnt);
        }
    }
    double p_out = 0.0, v_out = 0.0, a_out = 0.0;
    if (solved && comp) {
        // p = A_p a + A_initp [po;vo] ; v = A_v a + vo   (propStatedmpc.m:3-4).  The agent's state is read again here -- six SCALAR loads
        // through constant-address-space pointers (the values are wave-uniform) -- instead of living in registers across the solver loop
        typedef const double __attribute__((address_space(4))) *ConstD;
        const ConstD sp = (ConstD)(unsigned long long)(Qp->x_p + 3 * (size_t)gid), sv_ = (ConstD)(unsigned long long)(Qp->x_v + 3 * (size_t)gid);
        const double po0 = sp[0], po1 = sp[1], po2 = sp[2], vo0 = sv_[0], vo1 = sv_[1], vo2 = sv_[2];
        const double vo_l = ax_l == 0 ? vo0 : (ax_l == 1 ? vo1 : vo2);
        const double p0_l = init_pos(k_l, Qp->h, vo_l, ax_l == 0 ? po0 : (ax_l == 1 ? po1 : po2));   // A_initp(k,:) [po;vo]
        p_out = w + p0_l;
        v_out = vel_out(B + SL::A, k_l, ax_l, Qp->h, vo_l);
        a_out = a;
    }
    if (solved) {
        const bool ob_check = !(var == VAR_ELLIP || var == VAR_SOFTALL || var == VAR_SOFTALL_C || var == VAR_SCP || var == VAR_CPP1 || cppv);   // solveQPv2 / solveQP / solveSoftDMPC[_c] / solveDMPC have no in-bounds test
        if (h1.x & 4) status |= ST_COLL;   // cpp: collision noticed at the first step, solution still returned
        if (ob_check) {   // is_inbounds.m:2-5 on p(:,1)
            const double tolb = 50e-3;
            bool bad = false;
            const double hi3 = lane == 0 ? Qp->pmax[0] : (lane == 1 ? Qp->pmax[1] : Qp->pmax[2]), lo3 = lane == 0 ? Qp->pmin[0] : (lane == 1 ? Qp->pmin[1] : Qp->pmin[2]);
            if (lane < 3) bad = !(p_out < hi3 + tolb) || !(p_out > lo3 - tolb);
            if (__any(bad)) status |= ST_OUTBOUND;
        }
    }
    if (comp) {
        Qp->p_out[(size_t)gid * N3 + lane] = p_out;
        Qp->v_out[(size_t)gid * N3 + lane] = v_out;
        Qp->a_out[(size_t)gid * N3 + lane] = a_out;
        if (Qp->lT_next) {
            // next table chunk [S][3K][C]: uniform 64-bit base, 32-bit per-lane offset; unsolved agents keep their old prediction
            const int Cq = Qp->C;
            // (mixed precision: Qp->lT is the fp32 table of the scan; the fp64 predictions of this chunk are in Qp->own_prev)
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
            inf[0] = viol_k; inf[1] = nrows_built; inf[2] = tries; inf[3] = (!solved && (status & ST_COLL)) ? 0 : ccase;   // (`coll` return: no QP, no cost case)
            inf[4] = iters_total; inf[5] = nslack; inf[6] = solved ? q : 0; inf[7] = maxq;
#ifdef DMPC_DEV_TRACE
            if (ph_on && Qp->dbg_cap >= 4) {
                PH(9 > 8 ? 4 : 4);
                double *d = Qp->dbg + (size_t)(Qp->dbg_cap - 3) * 8;
                for (int u = 0; u < 20; ++u) d[u] = (double)phv[u];
            }
            if (Qp->dbg_agent == -6) { inf[5] = h1.w; inf[3] = cost; }   // development: the scan's key word (with the feature bits of the DEV_TRACE scan) and the work estimate
            if (Qp->dbg_agent == -4) { inf[0] = dev_nfast; inf[1] = dev_rounds; inf[3] = dev_negdrops; inf[5] = dev_tbl; inf[6] = dev_gen; }   // development: crash statistics in place of the branch record
#endif
        }
    }
}
