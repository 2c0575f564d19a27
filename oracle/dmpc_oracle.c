/*
 * dmpc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See dmpc_oracle.h.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  The dense QP is assembled literally (slack variables, +-Lambda rows,
 * bounds as rows) and solved by a dense Goldfarb-Idnani dual active-set method.
 * Only preallocated all-zero rows (0 <= 0; solveSoftDMPCbound.m:68-72, CollConstrHardDMPC.m:4)
 * are omitted: they can never become active.
 */
#include "dmpc_oracle.h"

#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXK 32
#define GIVENS_TINY 1e-200

/* ------------------------------------------------------------------------------------------ */
/* a1-a3 model matrices                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* getPosMat.m:4-22 / dmpc_soft_bound.m:82-108 / getDeltaMat.m:1-9: literal recurrences
 * new_row = Aux*prev_row + add_b ; A_init = Aux*A_init. */
int orc_model_matrices(double h, int K, double *Lambda, double *Av, double *A0, double *Delta)
{
    if (K < 1 || K > MAXK) return -1;
    const int n = 3 * K;
    double Aux[6][6] = {{1, 0, 0, h, 0, 0}, {0, 1, 0, 0, h, 0}, {0, 0, 1, 0, 0, h},
                        {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1}};
    double b[6][3] = {{h * h / 2, 0, 0}, {0, h * h / 2, 0}, {0, 0, h * h / 2},
                      {h, 0, 0},         {0, h, 0},         {0, 0, h}};
    double *prev = (double *)calloc((size_t)6 * n, sizeof(double));
    double *cur = (double *)calloc((size_t)6 * n, sizeof(double));
    double Ainit[6][6], T[6][6];
    memset(Ainit, 0, sizeof(Ainit));
    for (int i = 0; i < 6; ++i) Ainit[i][i] = 1.0;
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < 6; ++t) s += Aux[i][t] * prev[t * n + j];
                cur[i * n + j] = s;
            }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 3; ++j) cur[i * n + 3 * k + j] += b[i][j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < n; ++j) {
                if (Lambda) Lambda[(3 * k + i) * n + j] = cur[i * n + j];
                if (Av) Av[(3 * k + i) * n + j] = cur[(3 + i) * n + j];
            }
        memcpy(prev, cur, sizeof(double) * 6 * n);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0.0;
                for (int t = 0; t < 6; ++t) s += Aux[i][t] * Ainit[t][j];
                T[i][j] = s;
            }
        memcpy(Ainit, T, sizeof(T));
        if (A0)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 6; ++j) A0[(3 * k + i) * 6 + j] = Ainit[i][j];
    }
    free(prev);
    free(cur);
    if (Delta) {
        memset(Delta, 0, sizeof(double) * n * n);
        for (int i = 0; i < 3; ++i) Delta[i * n + i] = 1.0; /* getDeltaMat.m:3 */
        for (int k = 1; k < K; ++k)                         /* getDeltaMat.m:4-8: [-I I] */
            for (int i = 0; i < 3; ++i) {
                Delta[(3 * k + i) * n + 3 * (k - 1) + i] = -1.0;
                Delta[(3 * k + i) * n + 3 * k + i] = 1.0;
            }
    }
    return 0;
}

/* initDMPC.m:1-13: p(:,i) = po + t_i*(pf-po)/10, t_i = (i-1)h ; v = a = 0 */
int orc_init_one(const double po[3], const double pf[3], double h, int K, double *p, double *v, double *a)
{
    for (int i = 0; i < K; ++i) {
        double t = (double)i * h;
        for (int d = 0; d < 3; ++d) {
            double diff = pf[d] - po[d];
            p[3 * i + d] = po[d] + 1 * t * diff / 10;
            v[3 * i + d] = 0.0;
            a[3 * i + d] = 0.0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* dense Goldfarb-Idnani                                                                      */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int n, m, q;
    double *J;   /* n x n, column-major: J[i + n*j] */
    double *R;   /* n x n, column-major upper triangular */
    double *dv, *z, *r, *u;
    int *act, *where;
} gi_t;

static void gi_free(gi_t *g)
{
    free(g->J); free(g->R); free(g->dv); free(g->z); free(g->r); free(g->u); free(g->act); free(g->where);
}

static void gi_drop(gi_t *g, int l)
{
    const int n = g->n;
    g->where[g->act[l]] = -1;
    for (int k = l; k < g->q - 1; ++k) {
        g->act[k] = g->act[k + 1];
        g->u[k] = g->u[k + 1];
        g->where[g->act[k]] = k;
        memcpy(&g->R[(size_t)n * k], &g->R[(size_t)n * (k + 1)], sizeof(double) * (size_t)(k + 2));
    }
    g->q--;
    for (int j = l; j < g->q; ++j) {
        double a = g->R[j + (size_t)n * j], b = g->R[j + 1 + (size_t)n * j];
        if (fabs(b) < GIVENS_TINY) { g->R[j + 1 + (size_t)n * j] = 0.0; continue; } /* denormal guard */
        double rr = hypot(a, b), c = a / rr, s = b / rr;
        g->R[j + (size_t)n * j] = rr;
        g->R[j + 1 + (size_t)n * j] = 0.0;
        for (int k = j + 1; k < g->q; ++k) {
            double t1 = g->R[j + (size_t)n * k], t2 = g->R[j + 1 + (size_t)n * k];
            g->R[j + (size_t)n * k] = c * t1 + s * t2;
            g->R[j + 1 + (size_t)n * k] = -s * t1 + c * t2;
        }
        double *J1 = &g->J[(size_t)n * j], *J2 = &g->J[(size_t)n * (j + 1)];
        for (int i = 0; i < n; ++i) {
            double t1 = J1[i], t2 = J2[i];
            J1[i] = c * t1 + s * t2;
            J2[i] = -s * t1 + c * t2;
        }
    }
}

static void gi_add(gi_t *g, int p, double up)
{
    const int n = g->n, q = g->q;
    double *dv = g->dv;
    for (int j = n - 1; j > q; --j) {
        double a = dv[j - 1], b = dv[j];
        if (fabs(b) < GIVENS_TINY) { dv[j] = 0.0; continue; } /* denormal guard: c,s from denormals are garbage */
        double rr = hypot(a, b), c = a / rr, s = b / rr;
        dv[j - 1] = rr;
        dv[j] = 0.0;
        double *J1 = &g->J[(size_t)n * (j - 1)], *J2 = &g->J[(size_t)n * j];
        for (int i = 0; i < n; ++i) {
            double t1 = J1[i], t2 = J2[i];
            J1[i] = c * t1 + s * t2;
            J2[i] = -s * t1 + c * t2;
        }
    }
    for (int i = 0; i <= q; ++i) g->R[i + (size_t)n * q] = dv[i];
    g->act[q] = p;
    g->u[q] = up;
    g->where[p] = q;
    g->q = q + 1;
}

/* min 1/2 x'Gx + g'x s.t. C x <= d.  J0 = L^{-T} with G = L L' (n x n col-major, consumed);
 * x: in = unconstrained minimiser, out = solution.  Returns 0 ok, 1 infeasible, 2 iteration cap. */
static int gi_core(int n, int m, double *J0, double *x, const double *C, const double *d, double *lam,
                   int *iters_out, int *nact_out)
{
    gi_t g;
    g.n = n; g.m = m; g.q = 0;
    g.J = J0;
    g.R = (double *)calloc((size_t)n * n, sizeof(double));
    g.dv = (double *)malloc(sizeof(double) * n);
    g.z = (double *)malloc(sizeof(double) * n);
    g.r = (double *)malloc(sizeof(double) * n);
    g.u = (double *)malloc(sizeof(double) * n);
    g.act = (int *)malloc(sizeof(int) * n);
    g.where = (int *)malloc(sizeof(int) * (m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) g.where[i] = -1;
    const double tol = 1e-11;
    int iters = 0, rc = 0;
    const int itcap = 20 * (n + m) + 100;
    for (;;) {
        int p = -1;
        double worst = tol;
        for (int i = 0; i < m; ++i) {
            if (g.where[i] >= 0) continue;
            const double *ci = &C[(size_t)i * n];
            double s = -d[i];
            for (int j = 0; j < n; ++j) s += ci[j] * x[j];
            if (s > worst) { worst = s; p = i; }
        }
        if (p < 0) break;
        const double *cp = &C[(size_t)p * n];
        double up = 0.0;
        for (;;) {
            if (++iters > itcap) { rc = 2; goto done; }
            double dall = 0.0, dz = 0.0;
            for (int j = 0; j < n; ++j) {
                const double *Jj = &g.J[(size_t)n * j];
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += Jj[i] * cp[i];
                g.dv[j] = s;
                dall += s * s;
                if (j >= g.q) dz += s * s;
            }
            for (int i = 0; i < n; ++i) g.z[i] = 0.0;
            for (int j = g.q; j < n; ++j) {
                const double *Jj = &g.J[(size_t)n * j];
                double dj = g.dv[j];
                for (int i = 0; i < n; ++i) g.z[i] += Jj[i] * dj;
            }
            for (int i = g.q - 1; i >= 0; --i) { /* r = R^{-1} d1 */
                double s = g.dv[i];
                for (int k = i + 1; k < g.q; ++k) s -= g.R[i + (size_t)n * k] * g.r[k];
                g.r[i] = s / g.R[i + (size_t)n * i];
            }
            double viol = -d[p];
            for (int j = 0; j < n; ++j) viol += cp[j] * x[j];
            int dependent = !(dz > 1e-22 * dall) || g.q >= n;
            double t2 = dependent ? INFINITY : viol / dz; /* c_p'z = |J2' c_p|^2 = dz */
            double t1 = INFINITY;
            int l = -1;
            for (int k = 0; k < g.q; ++k)
                if (g.r[k] > 0.0) {
                    double t = g.u[k] / g.r[k];
                    if (t < t1) { t1 = t; l = k; }
                }
            double t = t1 < t2 ? t1 : t2;
            if (!(t < INFINITY)) { rc = 1; goto done; }
            if (dependent) { /* dual step only */
                for (int k = 0; k < g.q; ++k) g.u[k] -= t * g.r[k];
                up += t;
                gi_drop(&g, l);
                continue;
            }
            for (int i = 0; i < n; ++i) x[i] -= t * g.z[i];
            for (int k = 0; k < g.q; ++k) g.u[k] -= t * g.r[k];
            up += t;
            if (t2 <= t1) { gi_add(&g, p, up); break; }
            g.u[l] = 0.0;
            gi_drop(&g, l);
        }
    }
done:
    for (int i = 0; i < m; ++i) lam[i] = 0.0;
    for (int k = 0; k < g.q; ++k) lam[g.act[k]] = g.u[k];
    if (iters_out) *iters_out = iters;
    if (nact_out) *nact_out = g.q;
    gi_free(&g); /* frees J0 as g.J */
    return rc;
}

/* Cholesky G = L L' (row-major n x n in, lower L out in same storage order). returns 0 ok */
static int chol_lower(int n, const double *G, double *L)
{
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double s = G[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return -1;
        double ljj = sqrt(s);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = G[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / ljj;
        }
    }
    return 0;
}

/* Jt (row-major) = L^{-1}  (so J = L^{-T} has J[i + n*j] = Linv[j*n + i], i.e. column-major J
 * equals row-major Linv storage) */
static void invert_lower(int n, const double *L, double *Linv)
{
    memset(Linv, 0, sizeof(double) * n * n);
    for (int c = 0; c < n; ++c) {
        Linv[c * n + c] = 1.0 / L[c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += L[i * n + k] * Linv[k * n + c];
            Linv[i * n + c] = -s / L[i * n + i];
        }
    }
}

int orc_qp_dense(int n, int m, const double *G, const double *g, const double *C, const double *d,
                 double *x, double *lam, int *iters)
{
    double *L = (double *)malloc(sizeof(double) * n * n);
    double *Li = (double *)malloc(sizeof(double) * n * n);
    if (chol_lower(n, G, L)) { free(L); free(Li); return -2; }
    invert_lower(n, L, Li);
    /* column-major J[i + n*j] = (L^{-T})_{ij} = Linv_{ji} = Li[j*n+i]  -> same buffer */
    double *tmp = (double *)malloc(sizeof(double) * n);
    for (int j = 0; j < n; ++j) { /* tmp = J' g = Linv g */
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += Li[j * n + i] * g[i];
        tmp[j] = s;
    }
    for (int i = 0; i < n; ++i) { /* x = -J tmp */
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += Li[j * n + i] * tmp[j];
        x[i] = -s;
    }
    free(tmp);
    free(L);
    return gi_core(n, m, Li, x, C, d, lam, iters, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* per-run context: model matrices + cached Hessian factors per (q,s) cost case               */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    double q, s;
    double *H;    /* n3 x n3 row-major: 2(A'QA + Delta'S Delta + R) */
    double *Linv; /* L^{-1} row-major (== J col-major) */
    int valid;
} hcase_t;

typedef struct {
    int K, n3;
    double h;
    double *Lam, *Av, *A0, *Dl;
    hcase_t cases[4];
    pthread_mutex_t mu;
} ctx_t;

static ctx_t *ctx_new(double h, int K)
{
    ctx_t *c = (ctx_t *)calloc(1, sizeof(ctx_t));
    c->K = K; c->n3 = 3 * K; c->h = h;
    const int n = c->n3;
    c->Lam = (double *)malloc(sizeof(double) * n * n);
    c->Av = (double *)malloc(sizeof(double) * n * n);
    c->A0 = (double *)malloc(sizeof(double) * n * 6);
    c->Dl = (double *)malloc(sizeof(double) * n * n);
    orc_model_matrices(h, K, c->Lam, c->Av, c->A0, c->Dl);
    pthread_mutex_init(&c->mu, NULL);
    return c;
}

static void ctx_free(ctx_t *c)
{
    for (int i = 0; i < 4; ++i) { free(c->cases[i].H); free(c->cases[i].Linv); }
    free(c->Lam); free(c->Av); free(c->A0); free(c->Dl);
    pthread_mutex_destroy(&c->mu);
    free(c);
}

/* H = 2*(A'*Q*A + Delta'*S*Delta + R), Q = q*blkdiag(0,..,0,I3), S = s*I, R = I
 * (solveSoftDMPCbound.m:43-58,98), computed literally as dense triple products. */
static hcase_t *ctx_case(ctx_t *c, double q, double s)
{
    pthread_mutex_lock(&c->mu);
    hcase_t *hc = NULL;
    for (int i = 0; i < 4; ++i)
        if (c->cases[i].valid && c->cases[i].q == q && c->cases[i].s == s) { hc = &c->cases[i]; break; }
    if (!hc) {
        for (int i = 0; i < 4; ++i)
            if (!c->cases[i].valid) { hc = &c->cases[i]; break; }
        if (!hc) { hc = &c->cases[3]; free(hc->H); free(hc->Linv); }
        const int n = c->n3, K = c->K;
        double *QA = (double *)calloc((size_t)n * n, sizeof(double));
        for (int i = 3 * (K - 1); i < n; ++i) /* Q*A: only last 3 rows of Q nonzero */
            for (int j = 0; j < n; ++j) QA[i * n + j] = q * c->Lam[i * n + j];
        double *H = (double *)malloc(sizeof(double) * n * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double a = 0.0, b = 0.0;
                for (int t = 0; t < n; ++t) {
                    a += c->Lam[t * n + i] * QA[t * n + j];
                    b += c->Dl[t * n + i] * (s * c->Dl[t * n + j]);
                }
                H[i * n + j] = 2.0 * (a + b + (i == j ? 1.0 : 0.0));
            }
        free(QA);
        double *L = (double *)malloc(sizeof(double) * n * n);
        double *Li = (double *)malloc(sizeof(double) * n * n);
        chol_lower(n, H, L);
        invert_lower(n, L, Li);
        free(L);
        hc->q = q; hc->s = s; hc->H = H; hc->Linv = Li; hc->valid = 1;
    }
    pthread_mutex_unlock(&c->mu);
    return hc;
}

/* ------------------------------------------------------------------------------------------ */
/* a5/a6: scan + collision rows                                                               */
/* ------------------------------------------------------------------------------------------ */

/* ||E1 (p - pj)||_2 with E1 = diag(1,1,1/c)  (CheckCollSoftDMPC.m:10, order = 2) */
static double edist(const double *p, const double *pj, double c)
{
    double dx = p[0] - pj[0], dy = p[1] - pj[1], dz = (p[2] - pj[2]) / c;
    return sqrt(dx * dx + dy * dy + dz * dz);
}
/* norm(E1*(p - pj), order) for the super-ellipsoid of order 4 (CheckCollEllipDMPC.m:7, test/comp_test_ellipconstr.m:158-163) */
static double edist_o(const double *p, const double *pj, double c, int order)
{
    if (order == 2) return edist(p, pj, c);
    double dx = p[0] - pj[0], dy = p[1] - pj[1], dz = (p[2] - pj[2]) / c;
    return sqrt(sqrt(dx * dx * dx * dx + dy * dy * dy * dy + dz * dz * dz * dz));
}
/* order 4 exists for the variants whose scan is CheckCollEllipDMPC + rows for every neighbour (the one script of the reference that sets
 * order = 4, test/comp_test_ellipconstr.m:158-187, calls solveSoftDMPC) */
static int order_ok(const orc_params *prm)
{
    if (prm->order == 2) return 1;
    return prm->order == 4 && (prm->variant == ORC_SOFTALL || prm->variant == ORC_SOFTALL_C || prm->variant == ORC_ELLIP || prm->variant == ORC_REPAIR || prm->variant == ORC_CPP1);
}

typedef struct {
    int nrows;      /* number of collision rows */
    double *G;      /* nrows x n3: -xi' * Lambda[3kc-2:3kc,:]   (CollConstrSoftDMPC.m:27) */
    double *b;      /* rhs -r                                   (CollConstrSoftDMPC.m:28) */
    double *dist;   /* prev_dist                                (CollConstrSoftDMPC.m:19) */
    int cap;
} rows_t;

static void rows_push(rows_t *R, int n3)
{
    if (R->nrows == R->cap) {
        R->cap = R->cap ? 2 * R->cap : 64;
        R->G = (double *)realloc(R->G, sizeof(double) * (size_t)R->cap * n3);
        R->b = (double *)realloc(R->b, sizeof(double) * R->cap);
        R->dist = (double *)realloc(R->dist, sizeof(double) * R->cap);
    }
    R->nrows++;
}

/* one row, CollConstrSoftDMPC.m:16-28 (identical body in the Ellip/Hard/OnDemand/2 variants):
 *   dist = norm(E1*(p-pj(:,k)));  diff = (E2*(p-pj(:,k)))';
 *   r = dist*(rmin - dist + diff*p/dist) - diff*A_initp(3(kc-1)+1:3kc,:)*[po';vo'];
 *   Ain = -[0.. diff ..0]*A ; bin = -r
 * ke = step at which p, pj are evaluated (1-based), kc = step the row constrains (1-based). */
static void build_row(const ctx_t *c, const orc_params *prm, const double *l, int n, int j, int ke, int kc,
                      const double x0[6], rows_t *R)
{
    const int n3 = c->n3;
    const double *p = &l[(size_t)n * n3 + 3 * (ke - 1)];
    const double *pj = &l[(size_t)j * n3 + 3 * (ke - 1)];
    const double cc = prm->c;
    if (prm->order == 4) {
        /* CollConstrEllipDMPC.m:13-19 with order = 4, E1 = E^-1, E2 = E^-4 (comp_test_ellipconstr.m:160-163):
         *   dist = norm(E1*(p-pj),4); diff = (E2*(p-pj).^3)'; prev_dist = dist^3;
         *   r = dist^3*(rmin - dist + diff*p/dist^3) - diff*A_initp(3(kc-1)+1:3kc,:)*[po';vo'] */
        const double dist4 = edist_o(p, pj, cc, 4), pd = dist4 * dist4 * dist4, c4 = cc * cc * cc * cc;
        const double d0 = p[0] - pj[0], d1 = p[1] - pj[1], d2 = p[2] - pj[2];
        /* DMPC::solveQP's rows (dmpc/cpp/dmpc.cpp:47 _E2 = E1.^order, :478 diff = (_E2*(p - pj)).^(order-1)) scale BEFORE the power:
         * z component (c^-4 dz)^3, where the MATLAB helpers (CollConstrEllipDMPC.m:13, `.^` binds tighter) give c^-4 dz^3 */
        const double dzs = d2 / c4;
        const double df[3] = {d0 * d0 * d0, d1 * d1 * d1, prm->variant == ORC_CPP1 ? dzs * dzs * dzs : d2 * d2 * d2 / c4};
        const double dp4 = df[0] * p[0] + df[1] * p[1] + df[2] * p[2];
        double da4 = 0.0;
        for (int t = 0; t < 3; ++t) {
            const double *a0 = &c->A0[(size_t)(3 * (kc - 1) + t) * 6];
            double s = 0.0;
            for (int u = 0; u < 6; ++u) s += a0[u] * x0[u];
            da4 += df[t] * s;
        }
        const double r4 = pd * (prm->rmin - dist4 + dp4 / pd) - da4;
        rows_push(R, n3);
        double *g4 = &R->G[(size_t)(R->nrows - 1) * n3];
        for (int jj = 0; jj < n3; ++jj) {
            double s = 0.0;
            for (int t = 0; t < 3; ++t) s += df[t] * c->Lam[(size_t)(3 * (kc - 1) + t) * n3 + jj];
            g4[jj] = -s;
        }
        R->b[R->nrows - 1] = -r4;
        R->dist[R->nrows - 1] = pd;   /* prev_dist = dist^(order-1) */
        return;
    }
    double dist = edist(p, pj, cc);
    double diff[3] = {(p[0] - pj[0]), (p[1] - pj[1]), (p[2] - pj[2]) / (cc * cc)};
    double dp = diff[0] * p[0] + diff[1] * p[1] + diff[2] * p[2];
    double da = 0.0;
    for (int t = 0; t < 3; ++t) {
        const double *a0 = &c->A0[(size_t)(3 * (kc - 1) + t) * 6];
        double s = 0.0;
        for (int u = 0; u < 6; ++u) s += a0[u] * x0[u];
        da += diff[t] * s;
    }
    double r = dist * (prm->rmin - dist + dp / dist) - da;
    rows_push(R, n3);
    double *g = &R->G[(size_t)(R->nrows - 1) * n3];
    for (int jj = 0; jj < n3; ++jj) {
        double s = 0.0;
        for (int t = 0; t < 3; ++t) s += diff[t] * c->Lam[(size_t)(3 * (kc - 1) + t) * n3 + jj];
        g[jj] = -s;
    }
    R->b[R->nrows - 1] = -r;
    R->dist[R->nrows - 1] = dist;
}

typedef struct {
    int status;     /* ORC_ST_COLL or 0 */
    int viol_k;     /* 1-based first violating step handled (0 = none) */
    int nv;         /* number of selected neighbours */
    int nblocks;    /* number of row blocks (all3: up to 3, else 1; hard: K) */
    int rows_exist; /* ~isempty(Ain_coll) in the .m sense */
    int violation;  /* any(violation) / some_violation flag used for slack augmentation */
    int coll_flag;  /* cpp flavour: collision at the first step noticed, solve continues (dmpc.cpp:419-424) */
} scan_t;

/* dmpc.cpp:418  `_rmin*(1+(float)k/_k_hor)`: every operand is a float, so is the product (volatile: no contraction) */
static double cpp_near_radius(double rmin, int k0, int K)
{
    volatile float q = (float)k0 / (float)K;
    volatile float one_q = 1.0f + q;
    volatile float r = (float)rmin * one_q;
    return (double)r;
}

static int scp_check_coll(const double *p, const double *l, int N, int n, int k, int n3, double rmin);
static void scp_coll_constr(const ctx_t *c, const orc_params *prm, const double *p, const double *l, int N, int n, int k,
                            const double x0[6], rows_t *R);

/* solveSoftDMPCbound.m:21-38 and the corresponding loops of the other variants */
static scan_t scan_and_rows(const ctx_t *c, const orc_params *prm, int N, int n, const double *l,
                            const double x0[6], rows_t *R)
{
    scan_t sc;
    memset(&sc, 0, sizeof(sc));
    const int K = c->K, n3 = c->n3, var = prm->variant;
    const double rmin = prm->rmin;
    const double *own = &l[(size_t)n * n3];
    if (var == ORC_SCP) {
        /* the FIRST pass of solveDMPC.m:21-35 (prev_p = l(:,:,n), addConstr = []): rows for every other agent at the first violating step */
        for (int k = 1; k <= K; ++k)
            if (scp_check_coll(&own[3 * (k - 1)], l, N, n, k, n3, rmin)) {
                scp_coll_constr(c, prm, &own[3 * (k - 1)], l, N, n, k, x0, R);
                sc.viol_k = k; sc.nv = N - 1; sc.nblocks = 1; sc.rows_exist = R->nrows > 0; sc.violation = 1;
                break;
            }
        return sc;
    }
    if (var == ORC_HARD) {
        /* solveHardDMPC.m:18-22 + CollConstrHardDMPC.m:11-31: every k, every j with dist < 1 */
        for (int k = 1; k <= K; ++k)
            for (int j = 0; j < N; ++j) {
                if (j == n) continue;
                double d = edist(&own[3 * (k - 1)], &l[(size_t)j * n3 + 3 * (k - 1)], prm->c);
                if (d < 1) build_row(c, prm, l, n, j, k, k, x0, R);
            }
        sc.rows_exist = (N > 1); /* preallocated zero rows make Ain_coll non-empty */
        sc.nv = R->nrows;
        sc.nblocks = K;
        return sc;
    }
    const int cpp = (var == ORC_CPP || var == ORC_CPP2);
    const int soft_near = (var == ORC_BOUND || var == ORC_BOUND2 || var == ORC_ALL3 || var == ORC_ONDEMAND || cpp);
    const int coll_check = (var == ORC_BOUND || var == ORC_BOUND2 || var == ORC_ALL3 || var == ORC_REPAIR);
    const int skip_k1 = (var == ORC_BOUND2 || var == ORC_ALL3 || var == ORC_REPAIR || var == ORC_CPP2);
    /* DMPC::solveQP (dmpc.cpp:626-637): the first k (0-based) with check_collisions (:378-396: any neighbour with dist < _rmin) gets
     * rows for EVERY neighbour (build_collconstraint :450-498) on horizon step k-1 (`_A0.middleRows(3*(k-1),3)`, diff_row at 3*(k-1)).
     * k = 0 would index row -3 there: the reference's behaviour is undefined (Eigen assertion / out-of-bounds read); restated as the
     * `coll` outcome, no QP.  No first-step tolerance test, no near-neighbour selection. */
    const int cpp1 = (var == ORC_CPP1);
    unsigned char *sel = (unsigned char *)malloc((size_t)N);
    for (int k = 1; k <= K; ++k) {
        int any = 0, cnt = 0;
        double mind = INFINITY;
        for (int j = 0; j < N; ++j) {
            sel[j] = 0;
            if (j == n) continue;
            double d = edist_o(&own[3 * (k - 1)], &l[(size_t)j * n3 + 3 * (k - 1)], prm->c, prm->order);
            if (d < rmin) any = 1;               /* CheckCollSoftDMPC.m:11 */
            if (d < mind) mind = d;
            /* CheckCollSoftDMPC.m:12; cpp: viol_constr = dist < _rmin*(1+(float)k/_k_hor), float arithmetic, k 0-based
             * (dmpc.cpp:418; _rmin is a float member, dmpc.h:196) */
            const double near_r = cpp ? cpp_near_radius(rmin, k - 1, K) : rmin * (3);
            if (soft_near ? (d < near_r) : 1) { sel[j] = 1; cnt++; }
        }
        if (!any) continue;
        if (var == ORC_ALL3) sc.violation = 1; /* solveSoftDMPCall.m:22: some_violation set before the k==1 tests */
        /* cpp: `dist < _rmin - _collision_tol` (floats) only raises execution_ended, the build goes on (dmpc.cpp:419-424) */
        if (cpp && k == 1 && mind < (double)((float)rmin - 0.05f)) sc.coll_flag = 1;
        if (coll_check && k == 1 && mind < rmin - 0.05) { /* solveSoftDMPCbound.m:25-31 */
            sc.status = ORC_ST_COLL;
            sc.viol_k = 1;
            free(sel);
            return sc;
        }
        if (cpp1 && k == 1) { sc.status = ORC_ST_COLL; sc.viol_k = 1; free(sel); return sc; }
        if (skip_k1 && k == 1) continue; /* solveSoftDMPCbound2.m:29-31 */
        sc.viol_k = k;
        sc.nv = cnt;
        sc.violation = 1;
        sc.rows_exist = 1;
        if (var == ORC_ALL3) { /* solveSoftDMPCall.m:34-48 */
            int ks[3], nk = 0;
            if (k == 2) { ks[0] = k; ks[1] = k + 1; nk = 2; }
            else if (k == K) { ks[0] = k - 1; ks[1] = k; nk = 2; }
            else { ks[0] = k - 1; ks[1] = k; ks[2] = k + 1; nk = 3; }
            sc.nblocks = nk;
            for (int t = 0; t < nk; ++t)
                for (int j = 0; j < N; ++j)
                    if (sel[j]) build_row(c, prm, l, n, j, ks[t], ks[t], x0, R);
        } else {
            sc.nblocks = 1;
            int kc = (var == ORC_BOUND2 || var == ORC_CPP2 || cpp1) ? k - 1 : k; /* CollConstrSoftDMPC2.m:8; dmpc.cpp:516 k_ctr = k + _k_factor; solveQP: 3*(k-1), dmpc.cpp:480-485 */
            for (int j = 0; j < N; ++j)
                if (sel[j]) build_row(c, prm, l, n, j, k, kc, x0, R);
        }
        break;
    }
    free(sel);
    return sc;
}

/* ------------------------------------------------------------------------------------------ */
/* a7/a8: literal QP assembly + solve + a9 propagate + a10 in-bounds                          */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int has_slack;
    double coef_is_dist; /* 1: slack column diag(prev_dist); 0: eye */
    double lb;           /* slack lower bound (may be -INFINITY) */
    double lin;          /* linear cost term (per-row divided by dist if lin_over_dist) */
    int lin_over_dist;
    int ub_as_row;       /* solveSoftDMPC.m:21-23: eps <= 0 as extra rows (same maths) */
    double quad;         /* weight of eps^2 in the cost: EPS = quad * I (solveSoftDMPCbound.m:85-86: 1; solveSoftDMPC_c.m:63: 1e6 (K/k)^2) */
} slackcfg_t;

static slackcfg_t slack_cfg(const orc_params *prm)
{
    slackcfg_t s;
    memset(&s, 0, sizeof(s));
    s.quad = 1.0;
    switch (prm->variant) {
    case ORC_BOUND: s.has_slack = 1; s.coef_is_dist = 1; s.lb = -0.05; s.lin = prm->term; break;   /* :34,78,82 */
    case ORC_BOUND2: s.has_slack = 1; s.coef_is_dist = 1; s.lb = -0.01; s.lin = prm->term; break;  /* bound2:77 */
    case ORC_ALL3: s.has_slack = 1; s.coef_is_dist = 1; s.lb = -0.01; s.lin = prm->term; break;    /* all:92 */
    case ORC_SOFTALL: s.has_slack = 1; s.coef_is_dist = 0; s.lb = -INFINITY; s.lin = -1e5; s.ub_as_row = 1; break; /* solveSoftDMPC.m:21,65 */
    case ORC_SOFTALL_C: /* solveSoftDMPC_c.m:18-20,60-63: [Ainr I; 0 I], f_eps = -1e4 (K/k)^2, EPS = 1e6 (K/k)^2 -- k is only known after the scan: slack_cfg_c */
        s.has_slack = 1; s.coef_is_dist = 0; s.lb = -INFINITY; s.lin = -1e4; s.quad = 1e6; s.ub_as_row = 1; break;
    case ORC_CPP: case ORC_CPP2: /* dmpc.cpp:907-914: eps <= 0, -eps <= lim with float lim = 0.01 (:1079); term = -1e6 via params */
        s.has_slack = 1; s.coef_is_dist = 1; s.lb = -(double)0.01f; s.lin = prm->term; break;
    case ORC_CPP1: /* dmpc.cpp:629-633 [A_coll I; 0 I] x <= [b; 0]: coefficient 1, eps <= 0, no lower bound; f_w = -10^6 (:715), W = I (:719) */
        s.has_slack = 1; s.coef_is_dist = 0; s.lb = -INFINITY; s.lin = -1e6; s.ub_as_row = 1; break;
    case ORC_REPAIR: s.has_slack = 1; s.coef_is_dist = 1; s.lb = -INFINITY; s.lin = prm->term; s.lin_over_dist = 1; break; /* repair:33,77,81 */
    default: break;
    }
    return s;
}

/* solveSoftDMPC_c.m:60-63: `f_eps = -1*10^4*(K/k)^2*[...]`, `EPS = 1*10^6*(K/k)^2*[...]` with k the horizon step the scan broke at */
static void slack_cfg_c(slackcfg_t *s, const orc_params *prm, int K, int viol_k)
{
    if (prm->variant != ORC_SOFTALL_C || viol_k < 1) return;
    const double r = (double)K / (double)viol_k;
    s->lin = -1.0 * 1e4 * (r * r);
    s->quad = 1.0 * 1e6 * (r * r);
}

typedef struct {
    int n, m, ns;
    double *C, *d;  /* m x n, m */
    double *f;      /* n */
} qp_t;

static void qp_free(qp_t *q) { free(q->C); free(q->d); free(q->f); }

/* Assemble the literal problem (solveSoftDMPCbound.m:60-98). lbs/lin are the *current* ladder values. */
static void assemble(const ctx_t *c, const orc_params *prm, const rows_t *R, const scan_t *sc,
                     const slackcfg_t *sl, double qw, double sw, double lbs, double lin,
                     const double x0[6], const double ao[3], const double pf[3], qp_t *qp)
{
    const int n3 = c->n3, K = c->K;
    const int ns = (sl->has_slack && sc->violation) ? R->nrows : 0;
    const int n = n3 + ns;
    int m = R->nrows + 2 * n3 + 2 * n3; /* coll + +-Lambda + accel bounds */
    if (ns) m += ns /* eps<=0 */ + (isfinite(lbs) ? ns : 0);
    qp->n = n; qp->ns = ns; qp->m = m;
    qp->C = (double *)calloc((size_t)m * n, sizeof(double));
    qp->d = (double *)calloc((size_t)m, sizeof(double));
    qp->f = (double *)calloc((size_t)n, sizeof(double));
    int r = 0;
    for (int i = 0; i < R->nrows; ++i, ++r) { /* [Ainr diag(prev_dist)] (:34) */
        memcpy(&qp->C[(size_t)r * n], &R->G[(size_t)i * n3], sizeof(double) * n3);
        if (ns) qp->C[(size_t)r * n + n3 + i] = sl->coef_is_dist ? R->dist[i] : 1.0;
        qp->d[r] = R->b[i];
    }
    double a0x0[3 * MAXK];
    for (int i = 0; i < n3; ++i) {
        double s = 0.0;
        for (int u = 0; u < 6; ++u) s += c->A0[(size_t)i * 6 + u] * x0[u];
        a0x0[i] = s;
    }
    for (int i = 0; i < n3; ++i, ++r) { /* A a <= repmat(pmax) - A_initp*x0 (:72) */
        memcpy(&qp->C[(size_t)r * n], &c->Lam[(size_t)i * n3], sizeof(double) * n3);
        qp->d[r] = prm->pmax[i % 3] - a0x0[i];
    }
    for (int i = 0; i < n3; ++i, ++r) { /* -A a <= -repmat(pmin) + A_initp*x0 */
        for (int j = 0; j < n3; ++j) qp->C[(size_t)r * n + j] = -c->Lam[(size_t)i * n3 + j];
        qp->d[r] = -prm->pmin[i % 3] + a0x0[i];
    }
    for (int i = 0; i < n3; ++i, ++r) { qp->C[(size_t)r * n + i] = 1.0; qp->d[r] = prm->alim; }  /* ub (:4) */
    for (int i = 0; i < n3; ++i, ++r) { qp->C[(size_t)r * n + i] = -1.0; qp->d[r] = prm->alim; } /* lb (:5) */
    if (ns) {
        for (int i = 0; i < ns; ++i, ++r) { qp->C[(size_t)r * n + n3 + i] = 1.0; qp->d[r] = 0.0; } /* ub eps (:77) */
        if (isfinite(lbs))
            for (int i = 0; i < ns; ++i, ++r) { qp->C[(size_t)r * n + n3 + i] = -1.0; qp->d[r] = -lbs; } /* lb eps (:78) */
    }
    /* f = -2*(pf_rep'*Q*A - (A_initp*x0)'*Q*A + ao_1*S*Delta) + f_eps   (:88/:93) */
    for (int j = 0; j < n3; ++j) {
        double t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int i = 3 * (K - 1); i < n3; ++i) { /* Q nonzero only on last block */
            t1 += pf[i % 3] * qw * c->Lam[(size_t)i * n3 + j];
            t2 += a0x0[i] * qw * c->Lam[(size_t)i * n3 + j];
        }
        for (int i = 0; i < 3; ++i) t3 += ao[i] * sw * c->Dl[(size_t)i * n3 + j]; /* ao_1 = [ao 0 ...] */
        qp->f[j] = -2.0 * (t1 - t2 + t3);
    }
    for (int i = 0; i < ns; ++i) qp->f[n3 + i] = sl->lin_over_dist ? lin / R->dist[i] : lin;
}

static int in_bounds(const double p[3], const double *pmin, const double *pmax)
{ /* is_inbounds.m:2-5 (strict inequalities, tol = 50e-3) */
    const double tol = 50e-3;
    int up = p[0] < pmax[0] + tol && p[1] < pmax[1] + tol && p[2] < pmax[2] + tol;
    int down = p[0] > pmin[0] - tol && p[1] > pmin[1] - tol && p[2] > pmin[2] - tol;
    return up && down;
}

static void cost_case(const orc_params *prm, const scan_t *sc, const double po[3], const double pf[3],
                      double *qw, double *sw, int *which)
{
    double dn = sqrt((po[0] - pf[0]) * (po[0] - pf[0]) + (po[1] - pf[1]) * (po[1] - pf[1]) +
                     (po[2] - pf[2]) * (po[2] - pf[2]));
    int far = (prm->variant == ORC_ELLIP) ? (dn > 1) : (dn >= 1); /* solveEllipDMPC.m:26 uses > */
    const double sfree = prm->Sfree > 0 ? prm->Sfree : 10;
    if (!sc->rows_exist && far) { *qw = prm->Qfar > 0 ? prm->Qfar : 1000; *sw = sfree; *which = 0; }          /* :43-47 */
    else if (!sc->rows_exist && dn < 1) { *qw = prm->Qnear > 0 ? prm->Qnear : 10000; *sw = sfree; *which = 1; } /* :48-52 */
    else { *qw = prm->Q1; *sw = (prm->variant == ORC_ALL3) ? 10 : prm->S1; *which = 2; } /* :53-57; all:71 */
}

/* ------------------------------------------------------------------------------------------ */
/* solveDMPC.m:1-74 -- the legacy spherical SCP loop                                          */
/* ------------------------------------------------------------------------------------------ */

/* CheckCollDMPC.m:1-10: any other agent i with norm(p - l(:,k,i)) < r_min (plain Euclidean norm: no E1) */
static int scp_check_coll(const double *p, const double *l, int N, int n, int k, int n3, double rmin)
{
    int violation = 0;
    for (int i = 0; i < N; ++i) {
        if (i == n) continue;
        const double *pj = &l[(size_t)i * n3 + 3 * (k - 1)];
        const double dist = sqrt((p[0] - pj[0]) * (p[0] - pj[0]) + (p[1] - pj[1]) * (p[1] - pj[1]) + (p[2] - pj[2]) * (p[2] - pj[2]));
        violation = violation || (dist < rmin);
    }
    return violation;
}

/* CollConstrDMPC.m:1-34: one row per other agent i at horizon step k, linearised about the GIVEN p (prev_p(:,k) of the SCP loop):
 *   dist = norm(p - pj(:,k));  diff = (p - pj(:,k))';
 *   r = dist*(r_min - dist + (p - pj(:,k))'*p/dist) - (p - pj(:,k))'*A_initp(3(k-1)+1:3k,:)*[po';vo'];      (:16)
 *   Ain_total = [Ain_total; -diff_mat*Ain];  bin_total = [bin_total; -r]                                  (:23-24) */
static void scp_coll_constr(const ctx_t *c, const orc_params *prm, const double *p, const double *l, int N, int n, int k,
                            const double x0[6], rows_t *R)
{
    const int n3 = c->n3;
    for (int i = 0; i < N; ++i) {
        if (i == n) continue;
        const double *pj = &l[(size_t)i * n3 + 3 * (k - 1)];
        const double diff[3] = {p[0] - pj[0], p[1] - pj[1], p[2] - pj[2]};
        const double dist = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
        const double dp = diff[0] * p[0] + diff[1] * p[1] + diff[2] * p[2];
        double da = 0.0;
        for (int t = 0; t < 3; ++t) {
            const double *a0 = &c->A0[(size_t)(3 * (k - 1) + t) * 6];
            double s = 0.0;
            for (int u = 0; u < 6; ++u) s += a0[u] * x0[u];
            da += diff[t] * s;
        }
        const double r = dist * (prm->rmin - dist + dp / dist) - da;
        rows_push(R, n3);
        double *g = &R->G[(size_t)(R->nrows - 1) * n3];
        for (int jj = 0; jj < n3; ++jj) {
            double s = 0.0;
            for (int t = 0; t < 3; ++t) s += diff[t] * c->Lam[(size_t)(3 * (k - 1) + t) * n3 + jj];
            g[jj] = -s;
        }
        R->b[R->nrows - 1] = -r;
        R->dist[R->nrows - 1] = dist;
    }
}

/* maxDeviation.m:1-11.  `K = length(p)/3` is taken of the 3 x k_hor MATRIX p: length() = max(3, k_hor), so the loop `for k = 1:K` visits
 * only the first max(3, k_hor)/3 horizon steps (5 of 15) -- restated as written. */
static double scp_max_deviation(const double *p, const double *prev_p, int K)
{
    const int len = K > 3 ? K : 3;
    const int kmax = len / 3;   /* 1:K with K = len/3: k = 1 .. floor(K) */
    double tol = 0.0;           /* max() of an empty `dist` would be [] in MATLAB; k_hor >= 3 here */
    for (int k = 0; k < kmax; ++k) {
        const double d0 = p[3 * k] - prev_p[3 * k], d1 = p[3 * k + 1] - prev_p[3 * k + 1], d2 = p[3 * k + 2] - prev_p[3 * k + 2];
        const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if (k == 0 || dist > tol) tol = dist;
    }
    return tol;
}

/* solveDMPC.m:1-74.  `last`: optional export of the LAST pass's QP (certificates): its rows, cost weights and multipliers.
 * info: VIOLK = smallest member of addConstr (0: none), NV = |addConstr|, TRIES = SCP passes made, CASE = 0 / 2 of the last pass,
 * ITERS = active-set iterations of all passes, NROWS = collision rows of the last pass. */
typedef struct { qp_t qp; double qw, sw; int valid; } scp_last_t;
static int solve_scp(ctx_t *c, const orc_params *prm, int N, int n, const double *l, const double po[3], const double vo[3],
                     const double ao[3], const double pf[3], double *p, double *v, double *a, int *info, double *obj, scp_last_t *last)
{
    const int n3 = c->n3, K = c->K;   /* k_hor = size(l,2) (:4) */
    double x0[6] = {po[0], po[1], po[2], vo[0], vo[1], vo[2]};
    const double tol = prm->tol;
    double val = tol + 2;             /* :5 */
    int i = 1;                        /* :8 */
    unsigned char in_add[MAXK + 1];   /* addConstr (:9) as a membership table */
    memset(in_add, 0, sizeof(in_add));
    double prev_p[3 * MAXK], pn[3 * MAXK], vn[3 * MAXK], an[3 * MAXK];
    memcpy(prev_p, &l[(size_t)n * n3], sizeof(double) * n3);   /* prev_p = l(:,:,n) (:10) */
    slackcfg_t sl;
    memset(&sl, 0, sizeof(sl));
    sl.quad = 1.0;
    int status = ORC_ST_INFEAS, which = 0, nrows_last = 0;
    while (i <= K && val > tol) {     /* :17 */
        int newConstrCount = 0;       /* :18 */
        rows_t R;
        memset(&R, 0, sizeof(R));
        for (int k = 1; k <= K; ++k) {   /* :21-35 */
            const int violation = scp_check_coll(&prev_p[3 * (k - 1)], l, N, n, k, n3, prm->rmin);
            if (in_add[k]) scp_coll_constr(c, prm, &prev_p[3 * (k - 1)], l, N, n, k, x0, &R);
            else if (newConstrCount == 0 && violation) {
                scp_coll_constr(c, prm, &prev_p[3 * (k - 1)], l, N, n, k, x0, &R);
                in_add[k] = 1;
                newConstrCount = newConstrCount + 1;
            }
        }
        /* :38-48: isempty(Ain_total) -> Q = 1000, S = 10; else Q1, S1 */
        double qw, sw;
        if (R.nrows == 0) { qw = prm->Qfar > 0 ? prm->Qfar : 1000; sw = prm->Sfree > 0 ? prm->Sfree : 10; which = 0; }
        else { qw = prm->Q1; sw = prm->S1; which = 2; }
        hcase_t *hc = ctx_case(c, qw, sw);
        scan_t sc;
        memset(&sc, 0, sizeof(sc));
        qp_t qp;
        assemble(c, prm, &R, &sc, &sl, qw, sw, 0.0, 0.0, x0, ao, pf, &qp);   /* :50-54: [rows; A; -A], lb/ub, H, f */
        nrows_last = R.nrows;
        double *J = (double *)calloc((size_t)n3 * n3, sizeof(double));
        for (int j = 0; j < n3; ++j)
            for (int ii = 0; ii < n3; ++ii) J[ii + (size_t)n3 * j] = hc->Linv[(size_t)j * n3 + ii];
        double x[3 * MAXK], tmp[3 * MAXK];
        for (int j = 0; j < n3; ++j) {
            double s = 0.0;
            for (int ii = 0; ii < n3; ++ii) s += J[ii + (size_t)n3 * j] * qp.f[ii];
            tmp[j] = s;
        }
        for (int ii = 0; ii < n3; ++ii) {
            double s = 0.0;
            for (int j = 0; j < n3; ++j) s += J[ii + (size_t)n3 * j] * tmp[j];
            x[ii] = -s;
        }
        double *lam = (double *)malloc(sizeof(double) * qp.m);
        int iters = 0, nact = 0;
        const int rc = gi_core(n3, qp.m, J, x, qp.C, qp.d, lam, &iters, &nact);   /* quadprog (:57) */
        free(lam);
        info[ORC_I_ITERS] += iters;
        double objv = 0.0;
        if (rc == 0)
            for (int ii = 0; ii < n3; ++ii) {
                double s = 0.0;
                for (int j = 0; j < n3; ++j) s += hc->H[(size_t)ii * n3 + j] * x[j];
                objv += 0.5 * x[ii] * s + qp.f[ii] * x[ii];
            }
        if (last) { if (last->valid) qp_free(&last->qp); last->qp = qp; last->qw = qw; last->sw = sw; last->valid = 1; }
        else qp_free(&qp);
        free(R.G); free(R.b); free(R.dist);
        if (rc != 0) {   /* :58-63: isempty(a) -> p = [], v = [], success = 0 */
            status = ORC_ST_INFEAS;
            break;
        }
        memcpy(an, x, sizeof(double) * n3);
        for (int ii = 0; ii < n3; ++ii) {   /* propStatedmpc (:65; the stale 4-argument call: p = A a + A_initp [po;vo], v = A_v a + vo) */
            double sp = 0.0, sv = 0.0;
            for (int j = 0; j < n3; ++j) {
                sp += c->Lam[(size_t)ii * n3 + j] * an[j];
                sv += c->Av[(size_t)ii * n3 + j] * an[j];
            }
            double s0 = 0.0;
            for (int u = 0; u < 6; ++u) s0 += c->A0[(size_t)ii * 6 + u] * x0[u];
            pn[ii] = sp + s0;
            vn[ii] = sv + vo[ii % 3];
        }
        if (obj) *obj = objv;
        info[ORC_I_NACTIVE] = nact;
        status = ORC_ST_SOLVED;
        val = scp_max_deviation(pn, prev_p, K);   /* :69 */
        memcpy(prev_p, pn, sizeof(double) * n3);  /* :70 */
        i = i + 1;                                /* :71 */
    }
    int nadd = 0, kmin = 0;
    for (int k = K; k >= 1; --k) if (in_add[k]) { nadd++; kmin = k; }
    info[ORC_I_VIOLK] = kmin;
    info[ORC_I_NV] = nadd;
    info[ORC_I_TRIES] = status == ORC_ST_SOLVED ? i - 1 : i;
    info[ORC_I_CASE] = which;
    info[ORC_I_NROWS] = nrows_last;
    if (status == ORC_ST_SOLVED) {
        memcpy(p, pn, sizeof(double) * n3); memcpy(v, vn, sizeof(double) * n3); memcpy(a, an, sizeof(double) * n3);
    }
    return status;
}

static int solve_ctx(ctx_t *c, const orc_params *prm, int N, int n, const double *l, const double po[3],
                     const double vo[3], const double ao[3], const double pf[3], double *p, double *v,
                     double *a, int *info, double *obj)
{
    const int n3 = c->n3;
    int dummy[ORC_INFO_LEN];
    if (!info) info = dummy;
    memset(info, 0, sizeof(int) * ORC_INFO_LEN);
    if (prm->variant == ORC_SCP) return solve_scp(c, prm, N, n, l, po, vo, ao, pf, p, v, a, info, obj, NULL);
    double x0[6] = {po[0], po[1], po[2], vo[0], vo[1], vo[2]};
    rows_t R;
    memset(&R, 0, sizeof(R));
    scan_t sc = scan_and_rows(c, prm, N, n, l, x0, &R);
    info[ORC_I_VIOLK] = sc.viol_k;
    info[ORC_I_NV] = sc.nv;
    info[ORC_I_NROWS] = R.nrows;
    if (sc.status & ORC_ST_COLL) { free(R.G); free(R.b); free(R.dist); return ORC_ST_COLL; }
    double qw, sw;
    int which;
    cost_case(prm, &sc, po, pf, &qw, &sw, &which);
    info[ORC_I_CASE] = which;
    hcase_t *hc = ctx_case(c, qw, sw);
    slackcfg_t sl = slack_cfg(prm);
    slack_cfg_c(&sl, prm, c->K, sc.viol_k);
    const int cppv = (prm->variant == ORC_CPP || prm->variant == ORC_CPP2);
    const int ladder = (prm->variant == ORC_BOUND || prm->variant == ORC_BOUND2 || prm->variant == ORC_ALL3 || cppv);
    /* cpp: one solve + `while (status && tries < 20)` retries (dmpc.cpp:1081) */
    int max_tries = prm->max_tries > 0 ? prm->max_tries : (prm->variant == ORC_REPAIR ? 10 : (cppv ? 21 : 30));
    double lbs = sl.lb, lin = sl.lin;
    int tries = 0, status = ORC_ST_INFEAS;
    while (tries < max_tries) {
        qp_t qp;
        assemble(c, prm, &R, &sc, &sl, qw, sw, lbs, lin, x0, ao, pf, &qp);
        const int nn = qp.n;
        /* J = blkdiag(L^{-T}, I/sqrt(2 quad)) : slack block of H is 2*EPS = 2 quad I (:85-86,98; quad = 1 but for solveSoftDMPC_c.m:63) */
        double *J = (double *)calloc((size_t)nn * nn, sizeof(double));
        for (int j = 0; j < n3; ++j)
            for (int i = 0; i < n3; ++i) J[i + (size_t)nn * j] = hc->Linv[(size_t)j * n3 + i];
        for (int i = n3; i < nn; ++i) J[i + (size_t)nn * i] = 1.0 / sqrt(2.0 * sl.quad);
        double *x = (double *)malloc(sizeof(double) * nn);
        double *tmp = (double *)malloc(sizeof(double) * nn);
        for (int j = 0; j < nn; ++j) {
            double s = 0.0;
            for (int i = 0; i < nn; ++i) s += J[i + (size_t)nn * j] * qp.f[i];
            tmp[j] = s;
        }
        for (int i = 0; i < nn; ++i) {
            double s = 0.0;
            for (int j = 0; j < nn; ++j) s += J[i + (size_t)nn * j] * tmp[j];
            x[i] = -s;
        }
        free(tmp);
        double *lam = (double *)malloc(sizeof(double) * qp.m);
        int iters = 0, nact = 0;
        int rc = gi_core(nn, qp.m, J, x, qp.C, qp.d, lam, &iters, &nact);
        info[ORC_I_ITERS] += iters;
        if (rc == 0) {
            memcpy(a, x, sizeof(double) * n3);
            /* propStatedmpc.m:3-4 */
            for (int i = 0; i < n3; ++i) {
                double sp = 0.0, sv = 0.0;
                for (int j = 0; j < n3; ++j) {
                    sp += c->Lam[(size_t)i * n3 + j] * a[j];
                    sv += c->Av[(size_t)i * n3 + j] * a[j];
                }
                double s0 = 0.0;
                for (int u = 0; u < 6; ++u) s0 += c->A0[(size_t)i * 6 + u] * x0[u];
                p[i] = sp + s0;
                v[i] = sv + vo[i % 3];
            }
            if (obj) {
                double o = 0.0;
                for (int i = 0; i < n3; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < n3; ++j) s += hc->H[(size_t)i * n3 + j] * x[j];
                    o += 0.5 * x[i] * s + qp.f[i] * x[i];
                }
                for (int i = n3; i < nn; ++i) o += sl.quad * x[i] * x[i] + qp.f[i] * x[i];
                *obj = o;
            }
            int nsl = 0;
            for (int i = n3; i < nn; ++i)
                if (x[i] < -1e-12) nsl++;
            info[ORC_I_NSLACK] = nsl;
            info[ORC_I_NACTIVE] = nact;
            status = ORC_ST_SOLVED;
            const int ob_check = !(prm->variant == ORC_ELLIP || prm->variant == ORC_SOFTALL || prm->variant == ORC_SOFTALL_C || cppv || prm->variant == ORC_CPP1); /* solveQPv2 / solveQP have no in-bounds test */
            if (sc.coll_flag) status |= ORC_ST_COLL;
            if (ob_check && !in_bounds(p, prm->pmin, prm->pmax)) status |= ORC_ST_OUTBOUND; /* :125-128 */
            free(x); free(lam); qp_free(&qp);
            tries++;
            break;
        }
        free(x); free(lam); qp_free(&qp);
        tries++;
        if (ladder && sc.violation) { /* :147-153 */
            lbs = 2 * lbs;
            lin = 2 * lin;
            continue;
        }
        /* no-violation case / non-ladder variants: the retried problem is identical for an exact
         * solver (only ConstraintTolerance changes, :140-146) -> stays infeasible */
        tries = ladder || prm->variant == ORC_REPAIR ? max_tries : tries;
        break;
    }
    info[ORC_I_TRIES] = tries;
    free(R.G); free(R.b); free(R.dist);
    return status;
}

int orc_solve_one(const orc_params *prm, int N, int n, const double *l, const double po[3],
                  const double vo[3], const double ao[3], const double pf[3], double *p, double *v,
                  double *a, int *info, double *obj)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK) return -1;
    ctx_t *c = ctx_new(prm->h, prm->K);
    int st = solve_ctx(c, prm, N, n, l, po, vo, ao, pf, p, v, a, info, obj);
    ctx_free(c);
    return st;
}

typedef struct {
    ctx_t *c;
    const orc_params *prm;
    int N, lo, hi;
    const double *l, *x_p, *x_v, *x_a, *pf;
    double *p, *v, *a, *obj;
    int *status, *info;
} work_t;

static void *worker(void *arg)
{
    work_t *w = (work_t *)arg;
    const int n3 = w->c->n3;
    for (int i = w->lo; i < w->hi; ++i)
        w->status[i] = solve_ctx(w->c, w->prm, w->N, i, w->l, &w->x_p[3 * i], &w->x_v[3 * i], &w->x_a[3 * i],
                                 &w->pf[3 * i], &w->p[(size_t)i * n3], &w->v[(size_t)i * n3],
                                 &w->a[(size_t)i * n3], w->info ? &w->info[(size_t)i * ORC_INFO_LEN] : NULL,
                                 w->obj ? &w->obj[i] : NULL);
    return NULL;
}

int orc_step(const orc_params *prm, int N, const double *l, const double *x_p, const double *x_v,
             const double *x_a, const double *pf, double *p, double *v, double *a, int *status,
             int *info, double *obj, int nthreads)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > N) nthreads = N > 0 ? N : 1;
    /* as in orc_step_scenes: the dense QP of every solve is allocated and freed; keep those blocks in the per-thread heap arenas
     * (mmap/munmap per solve serialises the threads on the process's memory-map lock: 128 threads gave 1.6 x of 8) */
    mallopt(M_MMAP_THRESHOLD, 1 << 28);
    mallopt(M_TRIM_THRESHOLD, 1 << 29);
    ctx_t *c = ctx_new(prm->h, prm->K);
    work_t *w = (work_t *)calloc(nthreads, sizeof(work_t));
    pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
    /* contiguous clusters: N/T each, first N mod T get one more (dmpc.cpp:1600-1625) */
    int base = N / nthreads, rem = N % nthreads, lo = 0;
    for (int t = 0; t < nthreads; ++t) {
        int cnt = base + (t < rem ? 1 : 0);
        work_t ww = {c, prm, N, lo, lo + cnt, l, x_p, x_v, x_a, pf, p, v, a, obj, status, info};
        w[t] = ww;
        lo += cnt;
    }
    if (nthreads == 1) worker(&w[0]);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, &w[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    free(w); free(th);
    ctx_free(c);
    return 0;
}

int orc_eval_one(const orc_params *prm, int N, int n, const double *l, const double po[3],
                 const double vo[3], const double ao[3], const double pf[3], const double *acc,
                 double *obj, double *maxviol)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK) return -1;
    ctx_t *c = ctx_new(prm->h, prm->K);
    const int n3 = c->n3;
    double x0[6] = {po[0], po[1], po[2], vo[0], vo[1], vo[2]};
    rows_t R;
    memset(&R, 0, sizeof(R));
    scan_t sc = scan_and_rows(c, prm, N, n, l, x0, &R);
    int rc = 0;
    if (sc.status & ORC_ST_COLL) { rc = ORC_ST_COLL; goto out; }
    {
        double qw, sw;
        int which;
        cost_case(prm, &sc, po, pf, &qw, &sw, &which);
        hcase_t *hc = ctx_case(c, qw, sw);
        slackcfg_t sl = slack_cfg(prm);
        qp_t qp;
        assemble(c, prm, &R, &sc, &sl, qw, sw, sl.lb, sl.lin, x0, ao, pf, &qp);
        double o = 0.0, mv = 0.0;
        for (int i = 0; i < n3; ++i) {
            double s = 0.0;
            for (int j = 0; j < n3; ++j) s += hc->H[(size_t)i * n3 + j] * acc[j];
            o += 0.5 * acc[i] * s + qp.f[i] * acc[i];
        }
        /* optimal slack per row given acc: eps = min(0, (b - g'a)/coef) clipped at lb */
        for (int i = 0; i < R.nrows; ++i) {
            double s = 0.0;
            for (int j = 0; j < n3; ++j) s += R.G[(size_t)i * n3 + j] * acc[j];
            double res = s - R.b[i]; /* > 0 : violated without slack */
            if (qp.ns) {
                double coef = sl.coef_is_dist ? R.dist[i] : 1.0;
                double eps = res > 0 ? -res / coef : 0.0;
                if (eps < sl.lb) { if (res + coef * sl.lb > mv) mv = res + coef * sl.lb; eps = sl.lb; }
                o += eps * eps + qp.f[n3 + i] * eps;
            } else if (res > mv) mv = res;
        }
        for (int r = R.nrows; r < R.nrows + 4 * n3; ++r) {
            double s = 0.0;
            for (int j = 0; j < n3; ++j) s += qp.C[(size_t)r * qp.n + j] * acc[j];
            if (s - qp.d[r] > mv) mv = s - qp.d[r];
        }
        if (obj) *obj = o;
        if (maxviol) *maxviol = mv;
        qp_free(&qp);
    }
out:
    free(R.G); free(R.b); free(R.dist);
    ctx_free(c);
    return rc;
}

/* a5/a6 standalone: the rows scan_and_rows builds for agent n, dense (G: nrows x 3K, b, dist) */
int orc_rows_one(const orc_params *prm, int N, int n, const double *l, const double po[3], const double vo[3],
                 int max_rows, double *G, double *b, double *dist, int *nrows, int *viol_k, int *status)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK) return -1;
    ctx_t *c = ctx_new(prm->h, prm->K);
    double x0[6] = {po[0], po[1], po[2], vo[0], vo[1], vo[2]};
    rows_t R;
    memset(&R, 0, sizeof(R));
    scan_t sc = scan_and_rows(c, prm, N, n, l, x0, &R);
    *nrows = R.nrows; *viol_k = sc.viol_k; *status = sc.status;
    int cnt = R.nrows < max_rows ? R.nrows : max_rows;
    if (cnt > 0) {
        memcpy(G, R.G, sizeof(double) * (size_t)cnt * c->n3);
        memcpy(b, R.b, sizeof(double) * cnt);
        memcpy(dist, R.dist, sizeof(double) * cnt);
    }
    free(R.G); free(R.b); free(R.dist);
    ctx_free(c);
    return 0;
}

/* The literal dense QP of agent n at retry-ladder level `level` (lb_eps and term doubled `level` times,
 * solveSoftDMPCbound.m:147-153), exported for solver-independent certificates in tests/ (KKT by NNLS, infeasibility by
 * an LP): min 1/2 x'Hx + f'x s.t. C x <= d, x = [a; eps].  First call with H == NULL to get the sizes (*n_out, *m_out).
 * H: n x n (acceleration block = 2(A'QA + Delta'S Delta + R), slack block 2 I, :85-86,98), C: m x n row-major.
 * Returns ORC_ST_COLL when the variant returns before building a QP, 0 otherwise. */
int orc_assemble_one(const orc_params *prm, int N, int n, const double *l, const double po[3], const double vo[3],
                     const double ao[3], const double pf[3], int level, int *n_out, int *m_out, int *ncoll_out,
                     double *H, double *f, double *C, double *d)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK) return -1;
    ctx_t *c = ctx_new(prm->h, prm->K);
    const int n3 = c->n3;
    if (prm->variant == ORC_SCP) {   /* the QP of the LAST pass of the SCP loop (the one whose minimiser solveDMPC returns, or that proved infeasible) */
        scp_last_t last;
        memset(&last, 0, sizeof(last));
        double pp[3 * MAXK], vv[3 * MAXK], aa[3 * MAXK];
        int inf[ORC_INFO_LEN];
        memset(inf, 0, sizeof(inf));
        (void)level;
        solve_scp(c, prm, N, n, l, po, vo, ao, pf, pp, vv, aa, inf, NULL, &last);
        *n_out = last.qp.n; *m_out = last.qp.m;
        if (ncoll_out) *ncoll_out = inf[ORC_I_NROWS];
        if (H) {
            hcase_t *hc = ctx_case(c, last.qw, last.sw);
            for (int i = 0; i < n3; ++i) memcpy(&H[(size_t)i * n3], &hc->H[(size_t)i * n3], sizeof(double) * n3);
            memcpy(f, last.qp.f, sizeof(double) * n3);
            memcpy(C, last.qp.C, sizeof(double) * (size_t)last.qp.m * n3);
            memcpy(d, last.qp.d, sizeof(double) * last.qp.m);
        }
        qp_free(&last.qp);
        ctx_free(c);
        return 0;
    }
    double x0[6] = {po[0], po[1], po[2], vo[0], vo[1], vo[2]};
    rows_t R;
    memset(&R, 0, sizeof(R));
    scan_t sc = scan_and_rows(c, prm, N, n, l, x0, &R);
    int rc = 0;
    if (sc.status & ORC_ST_COLL) { rc = ORC_ST_COLL; *n_out = 0; *m_out = 0; if (ncoll_out) *ncoll_out = 0; goto out; }
    {
        double qw, sw;
        int which;
        cost_case(prm, &sc, po, pf, &qw, &sw, &which);
        hcase_t *hc = ctx_case(c, qw, sw);
        slackcfg_t sl = slack_cfg(prm);
        slack_cfg_c(&sl, prm, c->K, sc.viol_k);
        qp_t qp;
        assemble(c, prm, &R, &sc, &sl, qw, sw, ldexp(sl.lb, level), ldexp(sl.lin, level), x0, ao, pf, &qp);
        *n_out = qp.n; *m_out = qp.m;
        if (ncoll_out) *ncoll_out = R.nrows;
        if (H) {
            memset(H, 0, sizeof(double) * (size_t)qp.n * qp.n);
            for (int i = 0; i < n3; ++i) memcpy(&H[(size_t)i * qp.n], &hc->H[(size_t)i * n3], sizeof(double) * n3);
            for (int i = n3; i < qp.n; ++i) H[(size_t)i * qp.n + i] = 2.0 * sl.quad;
            memcpy(f, qp.f, sizeof(double) * qp.n);
            memcpy(C, qp.C, sizeof(double) * (size_t)qp.m * qp.n);
            memcpy(d, qp.d, sizeof(double) * qp.m);
        }
        qp_free(&qp);
    }
out:
    free(R.G); free(R.b); free(R.dist);
    ctx_free(c);
    return rc;
}

/* CPU baseline helper (bench.py, cpu_baseline leg): S independent scenes of N agents each, SCENE-parallel -- thread t solves
 * scenes t, t + T, ... with its own context (the scenes of a batch are independent problems, so this is how a host would
 * run a Monte-Carlo batch; the reference's own agent-cluster threads inside one scene, dmpc.cpp:1600-1625, are orc_step's
 * nthreads).  Arrays are [S][N][...] as in orc_step. */
typedef struct {
    const orc_params *prm;
    int S, N, t, T;
    const double *l, *x_p, *x_v, *x_a, *pf;
    double *p, *v, *a;
    int *status, *info;
} swork_t;

static void *scene_worker(void *arg)
{
    swork_t *w = (swork_t *)arg;
    ctx_t *c = ctx_new(w->prm->h, w->prm->K);
    const int n3 = c->n3, N = w->N;
    for (int s = w->t; s < w->S; s += w->T) {
        const size_t o3 = (size_t)s * N * 3, o45 = (size_t)s * N * n3;
        for (int i = 0; i < N; ++i)
            w->status[(size_t)s * N + i] =
                solve_ctx(c, w->prm, N, i, w->l + o45, &w->x_p[o3 + 3 * i], &w->x_v[o3 + 3 * i], &w->x_a[o3 + 3 * i], &w->pf[o3 + 3 * i],
                          &w->p[o45 + (size_t)i * n3], &w->v[o45 + (size_t)i * n3], &w->a[o45 + (size_t)i * n3],
                          w->info ? &w->info[((size_t)s * N + i) * ORC_INFO_LEN] : NULL, NULL);
    }
    ctx_free(c);
    return NULL;
}

int orc_step_scenes(const orc_params *prm, int S, int N, const double *l, const double *x_p, const double *x_v, const double *x_a,
                    const double *pf, double *p, double *v, double *a, int *status, int *info, int nthreads)
{
    if (!order_ok(prm) || prm->K < 1 || prm->K > MAXK || S < 1) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > S) nthreads = S;
    /* the dense QP of every solve is allocated and freed (as the reference's scripts do): keep those blocks in the per-thread
     * heap arenas instead of mmap/munmap per solve, which serialises all threads on the process's memory-map lock */
    mallopt(M_MMAP_THRESHOLD, 1 << 28);
    mallopt(M_TRIM_THRESHOLD, 1 << 29);
    swork_t *w = (swork_t *)calloc(nthreads, sizeof(swork_t));
    pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; ++t) {
        swork_t ww = {prm, S, N, t, nthreads, l, x_p, x_v, x_a, pf, p, v, a, status, info};
        w[t] = ww;
    }
    if (nthreads == 1) scene_worker(&w[0]);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, scene_worker, &w[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    free(w); free(th);
    return 0;
}
