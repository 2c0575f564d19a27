/* rqp_proto.c -- CPU prototype (development tool, not product, not oracle) of the REDUCED dual active-set solver for the
 * one-step-row slack variants (solveSoftDMPCbound.m): acceleration bounds and slack bounds as FIXED VARIABLES, the soft rows eliminated
 * into a 3x3 penalty in w_kc-space, hard rows / walls / the entering constraint as a small bordered system.  Every equality-constrained
 * QP of the Goldfarb-Idnani iteration is solved FROM SCRATCH (tridiagonal + rank-1 Hessian per axis: parallel cyclic reduction +
 * Sherman-Morrison), so there is no factor to update and nothing drifts.  The HIP kernel (csrc/dmpc_rsolve.hip) follows this file.
 *
 * Problem (per agent):  variables a[3][K] (|a| <= alim), eps[nr] (slb <= eps <= 0)
 *   min  sum_ax 1/2 a_ax' H1 a_ax + f_ax' a_ax + sum_j eps_j^2 + st eps_j,   H1 = T3 + 2 q lK lK',  T3 = 2 s D'D + 2 I
 *   s.t. -xi_j . w_kc + sd_j eps_j <= b_j,   wlo <= w <= whi,   w_k = sum_i L[k][i] a_i,  L[k][i] = h^2 (k - i + 1/2), i <= k
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define K 15
#define NRMAX 256
#define MWALL 6
#define MH 12
#define MD 10

typedef struct {
    double h, alim, q, s, st, slb;
    double f[3][K], whi[3][K], wlo[3][K];
    int kc, nr;
    double xi[NRMAX][3], b[NRMAX], sd[NRMAX];
} prob_t;

typedef struct {
    int iters, eqps, partial, singular, maxhard, maxextra, crashdrops, fallback_reason;
} stats_t;

int rqp_debug = 0;
int rqp_jitter = 0;   /* development: random factors on the pivot scores (every path of the method must end at the same minimiser) */
static unsigned long long jit_state = 88172645463325252ull;
static double jit(void) { if (!rqp_jitter) return 1.0; jit_state ^= jit_state << 13; jit_state ^= jit_state >> 7; jit_state ^= jit_state << 17; return 0.25 + 3.75 * (double)(jit_state >> 11) / 9007199254740992.0; }
enum { E_BOUND = 0, E_WALL = 1, E_ROW = 2, E_PIN0 = 3, E_PINL = 4, E_NONE = 5 };
typedef struct { int ty, ax, k, sg, j; } ent_t;

typedef struct {
    int fix[3][K];                          /* 0 free, +-1 fixed at +-alim */
    int rowin[NRMAX], pin0[NRMAX], pinL[NRMAX];
    int nw, wax[MWALL], wk[MWALL], wsg[MWALL];
    double mu[3][K], lam[NRMAX], pi[NRMAX], rho[NRMAX], lw[MWALL];
    double a[3][K], eps[NRMAX];
} gst_t;

/* ---------------------------------------------------------------- per-axis reduced Hessian solves */
typedef struct {
    double al[4][K], ga[4][K], binv[K];     /* PCR multipliers per stride, final reciprocal diagonal */
    double u[K], kap;                        /* u = T3_FF^-1 lK_F, kap = 1 / (1 + 2 q lK_F'u) */
    int fixd[K];
} axf_t;

static double lK(const prob_t *P, int k) { return P->h * P->h * ((double)(K - 1 - k) + 0.5); }
static double Lrow(const prob_t *P, int kr, int k) { return k <= kr ? P->h * P->h * ((double)(kr - k) + 0.5) : 0.0; }

static void pcr_apply_raw(const axf_t *A, const double *rhs, double *y)
{
    double r[K], rn[K];
    for (int i = 0; i < K; ++i) r[i] = A->fixd[i] ? 0.0 : rhs[i];
    for (int st = 0, s = 1; st < 4; ++st, s *= 2) {
        for (int i = 0; i < K; ++i) {
            const double lo = i - s >= 0 ? r[i - s] : 0.0, hi = i + s < K ? r[i + s] : 0.0;
            rn[i] = r[i] + A->al[st][i] * lo + A->ga[st][i] * hi;
        }
        memcpy(r, rn, sizeof(r));
    }
    for (int i = 0; i < K; ++i) y[i] = A->fixd[i] ? 0.0 : r[i] * A->binv[i];
}

static void ax_setup(const prob_t *P, const int *fix, axf_t *A)
{
    double a[K], b[K], c[K], an[K], bn[K], cn[K];
    const double e = -2.0 * P->s;
    for (int i = 0; i < K; ++i) {
        A->fixd[i] = fix[i] != 0;
        b[i] = A->fixd[i] ? 1.0 : (i < K - 1 ? 4.0 * P->s + 2.0 : 2.0 * P->s + 2.0);
        a[i] = (i > 0 && !A->fixd[i] && !fix[i - 1]) ? e : 0.0;
        c[i] = (i < K - 1 && !A->fixd[i] && !fix[i + 1]) ? e : 0.0;
    }
    for (int st = 0, s = 1; st < 4; ++st, s *= 2) {
        for (int i = 0; i < K; ++i) {
            const double al = i - s >= 0 ? -a[i] / b[i - s] : 0.0, ga = i + s < K ? -c[i] / b[i + s] : 0.0;
            A->al[st][i] = al; A->ga[st][i] = ga;
            bn[i] = b[i] + (i - s >= 0 ? al * c[i - s] : 0.0) + (i + s < K ? ga * a[i + s] : 0.0);
            an[i] = i - s >= 0 ? al * a[i - s] : 0.0;
            cn[i] = i + s < K ? ga * c[i + s] : 0.0;
        }
        memcpy(a, an, sizeof(a)); memcpy(b, bn, sizeof(b)); memcpy(c, cn, sizeof(c));
    }
    for (int i = 0; i < K; ++i) A->binv[i] = 1.0 / b[i];
    double l[K];
    for (int i = 0; i < K; ++i) l[i] = lK(P, i);
    pcr_apply_raw(A, l, A->u);
    double d = 0.0;
    for (int i = 0; i < K; ++i) if (!A->fixd[i]) d += l[i] * A->u[i];
    A->kap = 1.0 / (1.0 + 2.0 * P->q * d);
}

/* z = H1_FF^-1 nu_F (0 on the fixed components) */
static void ax_solve(const prob_t *P, const axf_t *A, const double *nu, double *z)
{
    double y[K];
    pcr_apply_raw(A, nu, y);
    double d = 0.0;
    for (int i = 0; i < K; ++i) if (!A->fixd[i]) d += lK(P, i) * y[i];
    const double cf = 2.0 * P->q * A->kap * d;
    for (int i = 0; i < K; ++i) z[i] = A->fixd[i] ? 0.0 : y[i] - cf * A->u[i];
}

/* (H1 v)_i for every i */
static void ax_hmul(const prob_t *P, const double *v, double *out)
{
    double d = 0.0;
    for (int i = 0; i < K; ++i) d += lK(P, i) * v[i];
    for (int i = 0; i < K; ++i) {
        const double dg = i < K - 1 ? 4.0 * P->s + 2.0 : 2.0 * P->s + 2.0;
        double t = dg * v[i];
        if (i > 0) t += -2.0 * P->s * v[i - 1];
        if (i < K - 1) t += -2.0 * P->s * v[i + 1];
        out[i] = t + 2.0 * P->q * lK(P, i) * d;
    }
}

/* ---------------------------------------------------------------- small dense helpers */
static int gauss_solve(int n, double *A /* n x n row-major, destroyed */, double *B /* n x m */, int m)
{
    for (int k = 0; k < n; ++k) {
        int p = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (best < 1e-300) return 1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < m; ++j) { double t = B[k * m + j]; B[k * m + j] = B[p * m + j]; B[p * m + j] = t; }
        }
        const double inv = 1.0 / A[k * n + k];
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double fct = A[i * n + k] * inv;
            if (fct == 0.0) continue;
            for (int j = k; j < n; ++j) A[i * n + j] -= fct * A[k * n + j];
            for (int j = 0; j < m; ++j) B[i * m + j] -= fct * B[k * m + j];
        }
    }
    for (int i = 0; i < n; ++i) { const double inv = 1.0 / A[i * n + i]; for (int j = 0; j < m; ++j) B[i * m + j] *= inv; }
    return 0;
}

/* ---------------------------------------------------------------- the equality-constrained QP of a working set (+ entering constraint) */
typedef struct {
    int singular;
    double a[3][K], eps[NRMAX];
    double lam[NRMAX], pi[NRMAX], rho[NRMAX], mu[3][K], lw[MWALL];
    double lam_p;                /* multiplier of the entering constraint (its own: bound / wall / row: lam; pin: pi / rho) */
    double farkas;               /* singular: d_p - sum r_c d_c  (< 0: the combination proves infeasibility when nothing blocks) */
    int nhard, nextra;
} eqp_t;

/* n' H1^-1 n of the UNREDUCED Hessian: the scale of the dependence test (as in the product kernel: delta <= 1e-13 s_pp) */
static __thread double sc_bound[K], sc_wall[K];
static void scales_init(const prob_t *P)
{
    int nofix[K] = {0};
    axf_t A;
    ax_setup(P, nofix, &A);
    for (int k = 0; k < K; ++k) {
        double e[K] = {0}, z[K], l[K];
        e[k] = 1.0;
        ax_solve(P, &A, e, z);
        sc_bound[k] = z[k];
        for (int i = 0; i < K; ++i) l[i] = Lrow(P, k, i);
        ax_solve(P, &A, l, z);
        double t = 0.0;
        for (int i = 0; i < K; ++i) t += l[i] * z[i];
        sc_wall[k] = t;
    }
}

static void eqp_solve(const prob_t *P, const gst_t *G, ent_t en, eqp_t *R, stats_t *S)
{
    S->eqps++;
    memset(R, 0, sizeof(*R));
    const int kc = P->kc, nr = P->nr;
    /* fixed sets (an entering bound is NOT fixed here: it is the last hard constraint) */
    axf_t AX[3];
    for (int x = 0; x < 3; ++x) ax_setup(P, G->fix[x], &AX[x]);
    /* a0: minimiser over the free components with the fixed ones at their bounds */
    double a0[3][K];
    for (int x = 0; x < 3; ++x) {
        double ab[K], hb[K], nu[K];
        for (int i = 0; i < K; ++i) ab[i] = G->fix[x][i] * P->alim;
        ax_hmul(P, ab, hb);
        for (int i = 0; i < K; ++i) nu[i] = -P->f[x][i] - hb[i];
        ax_solve(P, &AX[x], nu, a0[x]);
        for (int i = 0; i < K; ++i) if (G->fix[x][i]) a0[x][i] = ab[i];
    }
    /* basis of the small space: 0..2 the axes' l_kc, then the walls of W, then an entering wall / bound */
    int D = 3, eax[MD], ek[MD], esg[MD], ety[MD];
    for (int w = 0; w < G->nw; ++w) { eax[D] = G->wax[w]; ek[D] = G->wk[w]; esg[D] = G->wsg[w]; ety[D] = E_WALL; ++D; }
    if (en.ty == E_BOUND || en.ty == E_WALL) { eax[D] = en.ax; ek[D] = en.k; esg[D] = en.sg; ety[D] = en.ty; ++D; }
    if (D - 3 > S->maxextra) S->maxextra = D - 3;
    double nb[MD][K], Y[MD][K];   /* a-space normals (on their axis) and H~ n */
    int bax[MD];
    for (int d = 0; d < D; ++d) {
        bax[d] = d < 3 ? d : eax[d];
        for (int i = 0; i < K; ++i) {
            if (d < 3) nb[d][i] = Lrow(P, kc, i);
            else if (ety[d] == E_WALL) nb[d][i] = esg[d] * Lrow(P, ek[d], i);
            else nb[d][i] = (i == ek[d]) ? (double)esg[d] : 0.0;
        }
        ax_solve(P, &AX[bax[d]], nb[d], Y[d]);
    }
    double Gm[MD][MD], u0[MD];
    for (int i = 0; i < D; ++i) {
        for (int j = 0; j < D; ++j) {
            double t = 0.0;
            if (bax[i] == bax[j]) for (int k = 0; k < K; ++k) t += nb[j][k] * Y[i][k];
            Gm[i][j] = t;
        }
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += nb[i][k] * a0[bax[i]][k];
        u0[i] = t;
    }
    /* soft rows -> M_s, m_s;  hard list (entering last) */
    double Ms[3][3] = {{0}}, ms[3] = {0};
    int hrow[MH], nh = 0;         /* hard entries: row index >= 0, or -(basis index) - 1 for an extra */
    double hd[MH];
    int ej = (en.ty == E_ROW || en.ty == E_PIN0 || en.ty == E_PINL) ? en.j : -1;
    if (ej >= 0 && !(G->rowin[ej] && (G->pin0[ej] || G->pinL[ej]))) ej = -1;   /* the entering row is soft by now (its pin gave way): no bordered constraint */
    for (int j = 0; j < nr; ++j) {
        if (!G->rowin[j] || j == ej) continue;
        if (G->pin0[j] || G->pinL[j]) {
            if (nh >= MH - 1) { R->singular = -1; return; }
            hrow[nh] = j; hd[nh] = P->b[j] - P->sd[j] * (G->pinL[j] ? P->slb : 0.0); ++nh;
        } else {
            const double al = 2.0 / (P->sd[j] * P->sd[j]), be = 2.0 * P->b[j] / (P->sd[j] * P->sd[j]) + P->st / P->sd[j];
            for (int x = 0; x < 3; ++x) { ms[x] += be * P->xi[j][x]; for (int y = 0; y < 3; ++y) Ms[x][y] += al * P->xi[j][x] * P->xi[j][y]; }
        }
    }
    for (int d = 3; d < D; ++d) {
        const int last = (d == D - 1) && (en.ty == E_BOUND || en.ty == E_WALL);
        if (last) continue;
        if (nh >= MH - 1) { R->singular = -1; return; }
        hrow[nh] = -d - 1;
        hd[nh] = esg[d] > 0 ? P->whi[eax[d]][ek[d]] : -P->wlo[eax[d]][ek[d]];
        ++nh;
    }
    int has_p = (en.ty == E_BOUND || en.ty == E_WALL || ej >= 0);
    if (has_p) {
        if (en.ty == E_BOUND) { hrow[nh] = -(D - 1) - 1; hd[nh] = P->alim; }
        else if (en.ty == E_WALL) { hrow[nh] = -(D - 1) - 1; hd[nh] = en.sg > 0 ? P->whi[en.ax][en.k] : -P->wlo[en.ax][en.k]; }
        else { hrow[nh] = ej; hd[nh] = P->b[ej] - P->sd[ej] * (G->pinL[ej] ? P->slb : 0.0); }
        ++nh;
    }
    if (nh > S->maxhard) S->maxhard = nh;
    R->nhard = nh; R->nextra = D - 3;
    if (rqp_debug) fprintf(stderr, "   EQP ent %d idx %d nh %d ne %d  w0 %.10e %.10e %.10e  g %.6e %.6e %.6e\n", en.ty, (en.ty == E_BOUND || en.ty == E_WALL) ? 16 * en.ax + en.k : en.j, nh, D - 3, u0[0], u0[1], u0[2], Gm[0][0], Gm[1][1], Gm[2][2]);
    /* C_h columns in u-space */
    double Ch[MD][MH];
    for (int c = 0; c < nh; ++c)
        for (int d = 0; d < D; ++d) {
            if (hrow[c] >= 0) Ch[d][c] = d < 3 ? -P->xi[hrow[c]][d] : 0.0;
            else Ch[d][c] = (d == -hrow[c] - 1) ? 1.0 : 0.0;
        }
    /* K_u = I + Gm Mbar;  [ubar | Pu] = K_u^-1 [u0 - Gm mbar | Gm] */
    double Ku[MD * MD], RH[MD * (MD + 1)];
    for (int i = 0; i < D; ++i) {
        for (int j = 0; j < D; ++j) {
            double t = (i == j) ? 1.0 : 0.0;
            if (j < 3) for (int k = 0; k < 3; ++k) t += Gm[i][k] * Ms[k][j];
            Ku[i * D + j] = t;
        }
        double t = u0[i];
        for (int k = 0; k < 3; ++k) t -= Gm[i][k] * ms[k];
        RH[i * (D + 1)] = t;
        for (int j = 0; j < D; ++j) RH[i * (D + 1) + 1 + j] = Gm[i][j];
    }
    if (gauss_solve(D, Ku, RH, D + 1)) { R->singular = -1; return; }
    double ubar[MD], Pu[MD][MD];
    for (int i = 0; i < D; ++i) { ubar[i] = RH[i * (D + 1)]; for (int j = 0; j < D; ++j) Pu[i][j] = RH[i * (D + 1) + 1 + j]; }
    /* S_h = C_h' Pu C_h, Cholesky with the entering constraint last */
    double Sh[MH][MH], PC[MD][MH], rh[MH];
    for (int d = 0; d < D; ++d) for (int c = 0; c < nh; ++c) { double t = 0.0; for (int e = 0; e < D; ++e) t += Pu[d][e] * Ch[e][c]; PC[d][c] = t; }
    for (int c = 0; c < nh; ++c) {
        for (int e = 0; e < nh; ++e) { double t = 0.0; for (int d = 0; d < D; ++d) t += Ch[d][c] * PC[d][e]; Sh[c][e] = t; }
        double t = -hd[c];
        for (int d = 0; d < D; ++d) t += Ch[d][c] * ubar[d];
        rh[c] = t;
    }
    double Lc[MH][MH];
    memset(Lc, 0, sizeof(Lc));
    int dep = 0;
    for (int c = 0; c < nh; ++c) {
        for (int e = 0; e <= c; ++e) {
            double t = Sh[c][e];
            for (int k = 0; k < e; ++k) t -= Lc[c][k] * Lc[e][k];
            if (e < c) Lc[c][e] = t / Lc[e][e];
            else {
                double scale = Sh[c][c];
                if (hrow[c] >= 0) scale = sc_wall[kc] * (P->xi[hrow[c]][0] * P->xi[hrow[c]][0] + P->xi[hrow[c]][1] * P->xi[hrow[c]][1] + P->xi[hrow[c]][2] * P->xi[hrow[c]][2]);
                else { const int d = -hrow[c] - 1; scale = ety[d] == E_WALL ? sc_wall[ek[d]] : sc_bound[ek[d]]; }
                if (rqp_debug) fprintf(stderr, "      pivot c=%d/%d hrow %d t %.3e Shcc %.3e scale %.3e\n", c, nh, hrow[c], t, Sh[c][c], scale);
                if (!(t > 1e-13 * scale)) {
                    if (c == nh - 1 && has_p) { dep = 1; Lc[c][c] = 0.0; }
                    else { if (rqp_debug) fprintf(stderr, "   W dependent at c=%d of %d: t %.3e Shcc %.3e\n", c, nh, t, Sh[c][c]); R->singular = -1; return; }   /* the working set itself is dependent: not a state this method should reach */
                } else Lc[c][c] = sqrt(t);
            }
        }
    }
    double lamh[MH];
    if (!dep) {
        double y[MH], u[MD];
        for (int c = 0; c < nh; ++c) { double t = rh[c]; for (int k = 0; k < c; ++k) t -= Lc[c][k] * y[k]; y[c] = t / Lc[c][c]; }
        for (int c = nh - 1; c >= 0; --c) { double t = y[c]; for (int k = c + 1; k < nh; ++k) t -= Lc[k][c] * lamh[k]; lamh[c] = t / Lc[c][c]; }
        for (int d = 0; d < D; ++d) { double t = ubar[d]; for (int c = 0; c < nh; ++c) t -= PC[d][c] * lamh[c]; u[d] = t; }
        /* a posteriori: the hard constraints must hold at the computed point; when they do not, the small system is numerically singular
         * (a pivot of 1e-11 of its scale passed the test above, the multipliers are 1e17 and the point is noise): the entering constraint is
         * treated as dependent */
        if (has_p) {
            double worst = 0.0;
            for (int c = 0; c < nh; ++c) { double t = -hd[c]; for (int d = 0; d < D; ++d) t += Ch[d][c] * u[d]; if (fabs(t) > worst) worst = fabs(t); }
            if (worst > 1e-9) { if (rqp_debug) fprintf(stderr, "      residual %.3e: numerically dependent\n", worst); dep = 1; }
        }
    }
    if (!dep) {
        double u[MD], th[MD];
        for (int d = 0; d < D; ++d) { double t = ubar[d]; for (int c = 0; c < nh; ++c) t -= PC[d][c] * lamh[c]; u[d] = t; }
        for (int d = 0; d < D; ++d) {
            double t = 0.0;
            if (d < 3) { t = ms[d]; for (int k = 0; k < 3; ++k) t += Ms[d][k] * u[k]; }
            for (int c = 0; c < nh; ++c) t += Ch[d][c] * lamh[c];
            th[d] = t;
        }
        if (rqp_debug) fprintf(stderr, "      regular: w %.10e %.10e %.10e\n", u[0], u[1], u[2]);
        for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) R->a[x][i] = a0[x][i];
        for (int d = 0; d < D; ++d) for (int i = 0; i < K; ++i) if (!G->fix[bax[d]][i]) R->a[bax[d]][i] -= th[d] * Y[d][i];
        /* rows */
        for (int j = 0; j < nr; ++j) {
            R->eps[j] = 0.0; R->lam[j] = 0.0; R->pi[j] = -P->st; R->rho[j] = 0.0;
            if (!G->rowin[j]) continue;
            const int hard = G->pin0[j] || G->pinL[j], low = G->pinL[j];
            if (!hard) {
                const double xu = P->xi[j][0] * u[0] + P->xi[j][1] * u[1] + P->xi[j][2] * u[2];
                R->eps[j] = (P->b[j] + xu) / P->sd[j];
                R->lam[j] = -(2.0 * R->eps[j] + P->st) / P->sd[j];
                R->pi[j] = 0.0;
            } else {
                R->eps[j] = low ? P->slb : 0.0;
                for (int c = 0; c < nh; ++c) if (hrow[c] == j) R->lam[j] = lamh[c];
                if (low) { R->pi[j] = 0.0; R->rho[j] = 2.0 * P->slb + P->st + P->sd[j] * R->lam[j]; }
                else R->pi[j] = -P->st - P->sd[j] * R->lam[j];
            }
        }
        int wi = 0;
        for (int c = 0; c < nh; ++c) if (hrow[c] < 0) {
            const int d = -hrow[c] - 1;
            if (c == nh - 1 && has_p && (en.ty == E_BOUND || en.ty == E_WALL)) R->lam_p = lamh[c];
            else { (void)d; R->lw[wi++] = lamh[c]; }
        }
        /* multipliers of the fixed components: mu = -sigma (H a + f + N_b theta) */
        for (int x = 0; x < 3; ++x) {
            double ha[K];
            ax_hmul(P, R->a[x], ha);
            for (int i = 0; i < K; ++i) {
                double g = ha[i] + P->f[x][i];
                for (int d = 0; d < D; ++d) if (bax[d] == x) g += th[d] * nb[d][i];
                R->mu[x][i] = G->fix[x][i] ? -G->fix[x][i] * g : 0.0;
            }
        }
        return;
    }
    /* dependent: rates per unit of the entering multiplier (x does not move) */
    R->singular = 1;
    S->singular++;
    double rr[MH];
    {
        const int m = nh - 1;
        double y[MH];
        for (int c = 0; c < m; ++c) { double t = Sh[c][m]; for (int k = 0; k < c; ++k) t -= Lc[c][k] * y[k]; y[c] = t / Lc[c][c]; }
        for (int c = m - 1; c >= 0; --c) { double t = y[c]; for (int k = c + 1; k < m; ++k) t -= Lc[k][c] * rr[k]; rr[c] = t / Lc[c][c]; }
    }
    /* dlam of the last hard entry per unit t, and of the others: -rr * dlam_last */
    double dl_last = 1.0;
    if (en.ty == E_PIN0) dl_last = -1.0 / P->sd[ej];
    if (en.ty == E_PINL) dl_last = 1.0 / P->sd[ej];
    double dlh[MH];
    for (int c = 0; c < nh - 1; ++c) dlh[c] = -rr[c] * dl_last;
    dlh[nh - 1] = dl_last;
    R->lam_p = 1.0;
    double fk = 0.0;
    for (int c = 0; c < nh; ++c) fk += dlh[c] * hd[c];
    /* the pins' own right-hand sides: eps_j <= 0 (d = 0), -eps_j <= -slb (d = -slb) */
    double th[MD];
    for (int d = 0; d < D; ++d) { double t = 0.0; for (int c = 0; c < nh; ++c) t += Ch[d][c] * dlh[c]; th[d] = t; }
    for (int j = 0; j < nr; ++j) { R->lam[j] = 0.0; R->pi[j] = 0.0; R->rho[j] = 0.0; }
    int wi = 0;
    for (int c = 0; c < nh; ++c) {
        if (hrow[c] >= 0) {
            const int j = hrow[c];
            R->lam[j] = dlh[c];
            const int low = G->pinL[j];
            if (low) { R->rho[j] = P->sd[j] * dlh[c]; }
            else { R->pi[j] = -P->sd[j] * dlh[c]; }
        } else if (!(c == nh - 1 && (en.ty == E_BOUND || en.ty == E_WALL))) R->lw[wi++] = dlh[c];
    }
    for (int x = 0; x < 3; ++x)
        for (int i = 0; i < K; ++i) {
            double g = 0.0;
            for (int d = 0; d < D; ++d) if (bax[d] == x) g += th[d] * nb[d][i];
            R->mu[x][i] = G->fix[x][i] ? -G->fix[x][i] * g : 0.0;
            if (G->fix[x][i]) fk += R->mu[x][i] * P->alim;
        }
    R->farkas = fk;
}

/* ---------------------------------------------------------------- one ladder level: 0 solved, 1 infeasible, 2 give up (fallback) */
static void drop_row(const prob_t *P, gst_t *G, int j)
{
    G->rowin[j] = 0; G->pin0[j] = 1; G->pinL[j] = 0; G->lam[j] = 0.0; G->pi[j] = -P->st; G->rho[j] = 0.0; G->eps[j] = 0.0;
}

static int solve_level(const prob_t *P, gst_t *G, stats_t *S, int iter_cap)
{
    const double tol = 1e-10;
    const int nr = P->nr;
    memset(G, 0, sizeof(*G));
    scales_init(P);
    for (int j = 0; j < nr; ++j) { G->pin0[j] = 1; G->pi[j] = -P->st; }
    static __thread eqp_t R;
    ent_t none = {E_NONE, 0, 0, 0, 0};
    eqp_solve(P, G, none, &R, S);
    /* crash start: every bound violated at the unconstrained minimiser is fixed; negative multipliers are freed again */
    int nfix = 0;
    for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i)
        if (fabs(R.a[x][i]) - P->alim > tol) { G->fix[x][i] = R.a[x][i] > 0 ? 1 : -1; ++nfix; }
    if (nfix) {
        for (int pass = 0; pass < 64; ++pass) {
            eqp_solve(P, G, none, &R, S);
            if (R.singular) { S->fallback_reason = 1; return 2; }
            int neg = 0;
            for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i)
                if (G->fix[x][i] && R.mu[x][i] < 0.0) { G->fix[x][i] = 0; ++neg; S->crashdrops++; }
            if (!neg) break;
        }
    }
    memcpy(G->a, R.a, sizeof(G->a)); memcpy(G->eps, R.eps, sizeof(double) * nr);
    memcpy(G->mu, R.mu, sizeof(G->mu));
    for (int it = 0; it < iter_cap; ++it) {
        /* most violated constraint (the kernel's rule: plain violation for bounds and walls, 4 v / |xi| for rows, sqrt(2) v for the slack bounds) */
        double best = 0.0; ent_t p = none;
        double w[3][K];
        for (int x = 0; x < 3; ++x) for (int k = 0; k < K; ++k) { double t = 0.0; for (int i = 0; i <= k; ++i) t += Lrow(P, k, i) * G->a[x][i]; w[x][k] = t; }
        for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) {
            if (!G->fix[x][i]) { const double v = fabs(G->a[x][i]) - P->alim, sj = v * jit(); if (v > tol && sj > best) { best = sj; p = (ent_t){E_BOUND, x, i, G->a[x][i] > 0 ? 1 : -1, 0}; } }
            int inw = 0;
            for (int q = 0; q < G->nw; ++q) if (G->wax[q] == x && G->wk[q] == i) inw = 1;
            if (!inw) {
                const double c2 = w[x][i] - P->whi[x][i], c3 = P->wlo[x][i] - w[x][i], v = c2 > c3 ? c2 : c3;
                { const double sj = v * jit(); if (v > tol && sj > best) { best = sj; p = (ent_t){E_WALL, x, i, c2 > c3 ? 1 : -1, 0}; } }
            }
        }
        for (int j = 0; j < nr; ++j) {
            const double xn = sqrt(P->xi[j][0] * P->xi[j][0] + P->xi[j][1] * P->xi[j][1] + P->xi[j][2] * P->xi[j][2]);
            if (!G->rowin[j]) {
                const double v = -(P->xi[j][0] * w[0][P->kc] + P->xi[j][1] * w[1][P->kc] + P->xi[j][2] * w[2][P->kc]) - P->b[j];
                { const double sj = 4.0 * v / xn * jit(); if (v > tol && sj > best) { best = sj; p = (ent_t){E_ROW, 0, 0, 0, j}; } }
            } else if (!G->pin0[j] && !G->pinL[j]) {
                const double e = G->eps[j], lo = P->slb - e;
                { const double sj = 1.4142135 * e * jit(); if (e > tol && sj > best) { best = sj; p = (ent_t){E_PIN0, 0, 0, 0, j}; } }
                { const double sj = 1.4142135 * lo * jit(); if (lo > tol && sj > best) { best = sj; p = (ent_t){E_PINL, 0, 0, 0, j}; } }
            }
        }
        if (p.ty == E_NONE) {
            if (rqp_debug) {
                unsigned long long mhi = 0, mlo = 0;
                for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) { if (G->fix[x][i] > 0) mhi |= 1ull << (16 * x + i); if (G->fix[x][i] < 0) mlo |= 1ull << (16 * x + i); }
                fprintf(stderr, "  FINAL fixed hi %llx lo %llx\n", mhi, mlo);
                for (int j = 0; j < nr; ++j) {
                    const double v = -(P->xi[j][0] * w[0][P->kc] + P->xi[j][1] * w[1][P->kc] + P->xi[j][2] * w[2][P->kc]) - P->b[j];
                    fprintf(stderr, "  row %d in %d pin0 %d pinL %d  v(eps=0) %.4e eps %.4e  v+sd*eps %.3e lam %.3e pi %.3e rho %.3e\n", j, G->rowin[j], G->pin0[j], G->pinL[j], v, G->eps[j], v + P->sd[j] * G->eps[j], G->lam[j], G->pi[j], G->rho[j]);
                }
            }
            return 0;
        }
        S->iters++;
        if (rqp_debug) fprintf(stderr, "it %d: enter ty %d ax %d k %d sg %d j %d score %.3e\n", it, p.ty, p.ax, p.k, p.sg, p.j, best);
        /* rows and pins join the working set at once with multiplier 0 (they stay "entering": their own multiplier does not block,
         * and while hard they are the LAST hard constraint of the small system); bounds and walls stay outside until their full step */
        if (p.ty == E_ROW) G->rowin[p.j] = 1;
        if (p.ty == E_PIN0) { G->pin0[p.j] = 1; G->pi[p.j] = 0.0; }
        if (p.ty == E_PINL) { G->pinL[p.j] = 1; G->rho[p.j] = 0.0; }
        double lam_p = 0.0;
        for (int inner = 0;; ++inner) {
            if (inner > 200) { S->fallback_reason = 2; return 2; }
            if (p.ty == E_WALL && G->nw >= MWALL - 1) { S->fallback_reason = 3; return 2; }
            eqp_solve(P, G, p, &R, S);
            if (rqp_debug) fprintf(stderr, "   eqp: singular %d nhard %d nextra %d\n", R.singular, R.nhard, R.nextra);
            if (R.singular < 0) { S->fallback_reason = 4; return 2; }
            double tau = R.singular ? INFINITY : 1.0;
            int bty = -1, bx = 0, bi = 0;   /* blocker: 0 mu, 1 lam row, 2 pi, 3 rho, 4 wall */
#define RT(cur_, new_, ty_, x_, i_) do { const double c__ = (cur_), n__ = (new_); \
            if (R.singular) { if (n__ < 0.0) { const double t__ = c__ / -n__; if (t__ < tau) { tau = t__; bty = ty_; bx = x_; bi = i_; } } } \
            else if (n__ < 0.0) { const double t__ = c__ / (c__ - n__); if (t__ < tau) { tau = t__; bty = ty_; bx = x_; bi = i_; } } } while (0)
            for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) if (G->fix[x][i]) RT(G->mu[x][i], R.mu[x][i], 0, x, i);
            for (int j = 0; j < nr; ++j) {
                if (!G->rowin[j]) continue;
                const int own = (p.ty >= E_ROW && p.ty <= E_PINL && p.j == j) ? p.ty : -1;
                if (own != E_ROW) RT(G->lam[j], R.lam[j], 1, 0, j);
                if (G->pin0[j] && own != E_PIN0) RT(G->pi[j], R.pi[j], 2, 0, j);
                if (G->pinL[j] && own != E_PINL) RT(G->rho[j], R.rho[j], 3, 0, j);
            }
            for (int q = 0; q < G->nw; ++q) RT(G->lw[q], R.lw[q], 4, 0, q);
#undef RT
            if (R.singular) {
                if (!(tau < INFINITY)) return (R.farkas < 0.0) ? 1 : (S->fallback_reason = 5, 2);
                for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) if (G->fix[x][i]) G->mu[x][i] += tau * R.mu[x][i];
                for (int j = 0; j < nr; ++j) if (G->rowin[j]) { G->lam[j] += tau * R.lam[j]; G->pi[j] += tau * R.pi[j]; G->rho[j] += tau * R.rho[j]; }
                for (int q = 0; q < G->nw; ++q) G->lw[q] += tau * R.lw[q];
                lam_p += tau;
            } else if (tau >= 1.0) {
                memcpy(G->a, R.a, sizeof(G->a)); memcpy(G->eps, R.eps, sizeof(double) * nr);
                memcpy(G->mu, R.mu, sizeof(G->mu));
                memcpy(G->lam, R.lam, sizeof(double) * nr); memcpy(G->pi, R.pi, sizeof(double) * nr); memcpy(G->rho, R.rho, sizeof(double) * nr);
                memcpy(G->lw, R.lw, sizeof(G->lw));
                if (p.ty == E_BOUND) { G->fix[p.ax][p.k] = p.sg; G->mu[p.ax][p.k] = R.lam_p; }
                else if (p.ty == E_WALL) { G->wax[G->nw] = p.ax; G->wk[G->nw] = p.k; G->wsg[G->nw] = p.sg; G->lw[G->nw] = R.lam_p; G->nw++; }
                break;
            } else {
                S->partial++;
                /* (the primal moves along: when the inner loop ends without a full step -- the entering pin's row left -- it is the minimiser of the new working set) */
                for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) G->a[x][i] += tau * (R.a[x][i] - G->a[x][i]);
                for (int j = 0; j < nr; ++j) G->eps[j] += tau * (R.eps[j] - G->eps[j]);
                for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) if (G->fix[x][i]) G->mu[x][i] += tau * (R.mu[x][i] - G->mu[x][i]);
                for (int j = 0; j < nr; ++j) if (G->rowin[j]) {
                    G->lam[j] += tau * (R.lam[j] - G->lam[j]); G->pi[j] += tau * (R.pi[j] - G->pi[j]); G->rho[j] += tau * (R.rho[j] - G->rho[j]);
                }
                for (int q = 0; q < G->nw; ++q) G->lw[q] += tau * (R.lw[q] - G->lw[q]);
                lam_p += tau * (R.lam_p - lam_p);
            }
            if (rqp_debug) fprintf(stderr, "   partial tau %.3e blocker ty %d x %d i %d\n", tau, bty, bx, bi);
            /* drop the blocker */
            if (bty == 0) { G->fix[bx][bi] = 0; G->mu[bx][bi] = 0.0; }
            else if (bty == 1) {
                drop_row(P, G, bi);
                if ((p.ty == E_PIN0 || p.ty == E_PINL) && p.j == bi) break;   /* the entering pin's row left: nothing to add */
            }
            else if (bty == 2) { G->pin0[bi] = 0; G->pi[bi] = 0.0; }
            else if (bty == 3) { G->pinL[bi] = 0; G->rho[bi] = 0.0; }
            else if (bty == 4) { for (int q = bi; q < G->nw - 1; ++q) { G->wax[q] = G->wax[q + 1]; G->wk[q] = G->wk[q + 1]; G->wsg[q] = G->wsg[q + 1]; G->lw[q] = G->lw[q + 1]; } G->nw--; }
        }
    }
    S->fallback_reason = 8;
    return 2;
}

/* entry: solves with the retry ladder; a_out[3][K] (axis-major), returns 0 solved / 1 infeasible after max_tries / 2 fallback; *tries */
int rqp_solve(const prob_t *P0, int max_tries, int iter_cap, double *a_out, int *tries_out, int *stats_out)
{
    prob_t P = *P0;
    static __thread gst_t G;
    stats_t S;
    memset(&S, 0, sizeof(S));
    int tries = 0, rc = 1;
    while (tries < max_tries) {
        ++tries;
        rc = solve_level(&P, &G, &S, iter_cap);
        if (rc != 1) break;
        if (P.nr == 0) { tries = max_tries; break; }
        P.slb *= 2.0; P.st *= 2.0;
    }
    if (rc == 0) for (int x = 0; x < 3; ++x) for (int i = 0; i < K; ++i) a_out[3 * i + x] = G.a[x][i];
    *tries_out = tries;
    if (stats_out) { stats_out[0] = S.iters; stats_out[1] = S.eqps; stats_out[2] = S.partial; stats_out[3] = S.singular; stats_out[4] = S.maxhard; stats_out[5] = S.maxextra; stats_out[6] = S.crashdrops; stats_out[7] = S.fallback_reason; }
    return rc;
}

int rqp_prob_size(void) { return (int)sizeof(prob_t); }
