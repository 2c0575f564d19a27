// dmpc_rowbuild.hip -- dense collision-row builders behind the reference's CollConstr*/AddCollConstr helpers
// (SURVEY.md §8 f-3).  These helpers return the DENSE inequality rows  Ain = -diff_mat * A,  bin = -r  that the
// sibling algorithms hand to quadprog (dec-iSCP/CollConstr.m:1-24, cup-SCP/AddCollConstr.m:1-31,
// dmpc/matlab/CollConstrSoftDMPC.m:1-32 and variants).  diff_mat has one non-zero 1x3 block (two for the pairwise
// cup-SCP rows), so every output element is a 3-term (6-term) dot product with rows of A: pure streaming work,
// bound by the HBM write of Ain.  A and the outputs carry explicit strides so MATLAB's column-major arrays and
// row-major hosts bind without a transpose.
//
// Included into dmpc_api.hip (single translation unit).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rb {

struct RowGeom { double xi[3]; double dist; double r; };

// The ellipsoid of the helpers for order 2 and 4 (CollConstr.m:10-14, CollConstrSoftDMPC.m:16-21, AddCollConstr.m:13-17; E1 = E^-1, E2 = E^-order,
// `.^` binds tighter than `*`):  dist = |E1 d|_order,  diff = E2 d.^(order-1),  pd = dist^(order-1).  Returns dist; xi = diff, pd by reference.
__device__ __forceinline__ double ell_geom(double dx, double dy, double dz, double cinv, int order, double xi[3], double &pd)
{
    const double ez = dz * cinv;
    if (order == 4) {
        const double dist = sqrt(sqrt(dx * dx * dx * dx + dy * dy * dy * dy + ez * ez * ez * ez));
        const double c2 = cinv * cinv;
        xi[0] = dx * dx * dx; xi[1] = dy * dy * dy; xi[2] = dz * dz * dz * (c2 * c2);
        pd = dist * dist * dist;
        return dist;
    }
    const double dist = sqrt(dx * dx + dy * dy + ez * ez);
    xi[0] = dx; xi[1] = dy; xi[2] = dz * cinv * cinv;
    pd = dist;
    return dist;
}

// r = pd (rmin - dist + diff.p / pd) - diff.a0;  g.dist = prev_dist = pd (what CollConstrSoftDMPC returns as its third output)
__device__ __forceinline__ RowGeom row_geom(const double p[3], const double pj[3], const double a0[3], double rmin, double cinv, int order)
{
    RowGeom g;
    double pd;
    const double dist = ell_geom(p[0] - pj[0], p[1] - pj[1], p[2] - pj[2], cinv, order, g.xi, pd);
    const double dp = g.xi[0] * p[0] + g.xi[1] * p[1] + g.xi[2] * p[2];
    const double da = g.xi[0] * a0[0] + g.xi[1] * a0[1] + g.xi[2] * a0[2];
    g.r = pd * (rmin - dist + dp / pd) - da;
    g.dist = pd;
    return g;
}

// rows against a list of obstacles at one time step: out[r][c] = -(xi_r . A[3 kb + (0..2)][c])
// grid: linear over n_sel * ncols elements in the output's fast-dimension order
__global__ void coll_rows_kernel(int n_sel, const int *__restrict__ sel, int K, const double *__restrict__ l, int k_cmp, int k_blk,
                                 double p0, double p1, double p2, double a00, double a01, double a02, double rmin, double cinv,
                                 const double *__restrict__ A, long a_rs, long a_cs, int ncols, double *__restrict__ Ain, long o_rs,
                                 long o_cs, double *__restrict__ bin, double *__restrict__ dist, int order)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_sel * ncols) return;
    int r, c;
    if (o_cs <= o_rs) { r = (int)(e / ncols); c = (int)(e - (size_t)r * ncols); }   // row-major output: columns fastest
    else { c = (int)(e / n_sel); r = (int)(e - (size_t)c * n_sel); }                // column-major output: rows fastest
    const double p[3] = {p0, p1, p2}, a0[3] = {a00, a01, a02};
    const double *pj = l + ((size_t)sel[r] * K + k_cmp) * 3;
    const double q[3] = {pj[0], pj[1], pj[2]};
    const RowGeom g = row_geom(p, q, a0, rmin, cinv, order);
    const double *Ab = A + (size_t)(3 * k_blk) * a_rs + (size_t)c * a_cs;
    Ain[(size_t)r * o_rs + (size_t)c * o_cs] = -(g.xi[0] * Ab[0] + g.xi[1] * Ab[a_rs] + g.xi[2] * Ab[2 * a_rs]);
    if (c == 0) { bin[r] = -g.r; if (dist) dist[r] = g.dist; }
}

// dense form of structured rows (xi_r, kc_r) as dmpc_rows_one returns them: out[r][c] = -(xi_r . A[3 (kc_r - 1) + (0..2)][c])
__global__ void xi_rows_kernel(int nr, const double *__restrict__ xi, const int *__restrict__ kc, const double *__restrict__ A,
                               long a_rs, long a_cs, int ncols, double *__restrict__ out, long o_rs, long o_cs)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nr * ncols) return;
    const int r = (int)(e / ncols), c = (int)(e - (size_t)r * ncols);
    const double *Ab = A + (size_t)(3 * (kc[r] - 1)) * a_rs + (size_t)c * a_cs;
    out[(size_t)r * o_rs + (size_t)c * o_cs] = -(xi[3 * r] * Ab[0] + xi[3 * r + 1] * Ab[a_rs] + xi[3 * r + 2] * Ab[2 * a_rs]);
}

// cup-SCP pairwise rows: row (i < j, k), k fastest (AddCollConstr.m:8-29):
//   r = dist (rmin - dist) + diff.(pi_k - pj_k) - diff.(po_i - po_j);  Ain = -(diff . A[blk(i,k)] - diff . A[blk(j,k)])
__global__ void add_coll_rows_kernel(int N, int K, const double *__restrict__ p, const double *__restrict__ po, double rmin,
                                     double cinv, const double *__restrict__ A, long a_rs, long a_cs, int ncols,
                                     double *__restrict__ Ain, long o_rs, long o_cs, double *__restrict__ bin, size_t nrows, int order)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nrows * (size_t)ncols) return;
    size_t r; int c;
    if (o_cs <= o_rs) { r = e / ncols; c = (int)(e - r * ncols); }
    else { c = (int)(e / nrows); r = e - (size_t)c * nrows; }
    const int k = (int)(r % K);
    const long pr = (long)(r / K);            // pair rank in (0,1),(0,2),...,(0,N-1),(1,2),...
    int i = (int)((2.0 * N - 1 - sqrt((2.0 * N - 1) * (2.0 * N - 1) - 8.0 * (double)pr)) / 2.0);
    while ((long)i * (2 * N - i - 1) / 2 > pr) --i;
    while ((long)(i + 1) * (2 * N - i - 2) / 2 <= pr) ++i;
    const int j = (int)(pr - (long)i * (2 * N - i - 1) / 2) + i + 1;
    const double *pi = p + ((size_t)i * K + k) * 3, *pj = p + ((size_t)j * K + k) * 3;
    const double dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
    double xi_[3], pd;
    const double dist = ell_geom(dx, dy, dz, cinv, order, xi_, pd);
    const double x0 = xi_[0], x1 = xi_[1], x2 = xi_[2];
    const double *Ai = A + (size_t)(3 * ((size_t)K * i + k)) * a_rs + (size_t)c * a_cs;
    const double *Aj = A + (size_t)(3 * ((size_t)K * j + k)) * a_rs + (size_t)c * a_cs;
    Ain[r * (size_t)o_rs + (size_t)c * o_cs] =
        -((x0 * Ai[0] + x1 * Ai[a_rs] + x2 * Ai[2 * a_rs]) - (x0 * Aj[0] + x1 * Aj[a_rs] + x2 * Aj[2 * a_rs]));
    if (c == 0) {
        const double *oi = po + (size_t)i * 3, *oj = po + (size_t)j * 3;
        const double rr = pd * (rmin - dist) + (x0 * dx + x1 * dy + x2 * dz) -
                          (x0 * (oi[0] - oj[0]) + x1 * (oi[1] - oj[1]) + x2 * (oi[2] - oj[2]));
        bin[r] = -rr;
    }
}


__device__ __forceinline__ void unrank_pair(long pr, int N, int &i, int &j)
{
    i = (int)((2.0 * N - 1 - sqrt((2.0 * N - 1) * (2.0 * N - 1) - 8.0 * (double)pr)) / 2.0);
    while ((long)i * (2 * N - i - 1) / 2 > pr) --i;
    while ((long)(i + 1) * (2 * N - i - 2) / 2 <= pr) ++i;
    j = (int)(pr - (long)i * (2 * N - i - 1) / 2) + i + 1;
}

__device__ __forceinline__ double pair_geom(const double *__restrict__ p, const double *__restrict__ po, int K, int i, int j, int k,
                                            double rmin, double cinv, double xi[3], int order)
{
    const double *pi = p + ((size_t)i * K + k) * 3, *pj = p + ((size_t)j * K + k) * 3;
    const double dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
    double pd;
    const double dist = ell_geom(dx, dy, dz, cinv, order, xi, pd);
    const double *oi = po + (size_t)i * 3, *oj = po + (size_t)j * 3;
    return pd * (rmin - dist) + (xi[0] * dx + xi[1] * dy + xi[2] * dz) -
           (xi[0] * (oi[0] - oj[0]) + xi[1] * (oi[1] - oj[1]) + xi[2] * (oi[2] - oj[2]));
}

// ROW-MAJOR output (o_cs == 1).  Block = (tile of 256 columns, horizon step k, chunk of RB_PCH pairs): for one k and one
// column, all N(N-1)/2 pair rows need only the 3N values A(blk(.,k), c), so the A slab of a tile stays in L1/L2 while
// the blocks stream out full-width row segments -- one HBM write and <= 3 cached reads per element instead of 6
// scattered reads.  The pair geometry of the block is computed once and shared through LDS.
#define RB_PCH 64
__global__ __launch_bounds__(256) void add_coll_rows_rm_kernel(int N, int K, const double *__restrict__ p,
                                                               const double *__restrict__ po, double rmin, double cinv,
                                                               const double *__restrict__ A, long a_rs, long a_cs, int ncols,
                                                               double *__restrict__ Ain, long o_rs, double *__restrict__ bin, int order)
{
    __shared__ double gx[RB_PCH][3];
    __shared__ int gij[RB_PCH][2];
    const int k = blockIdx.y, tid = threadIdx.x;
    const long npairs = (long)N * (N - 1) / 2;
    const long base = (long)blockIdx.z * RB_PCH;
    const int cnt = (int)min((long)RB_PCH, npairs - base);
    const int c = blockIdx.x * 256 + tid;
    if (tid < cnt) {
        int i, j;
        unrank_pair(base + tid, N, i, j);
        double xi[3];
        const double r = pair_geom(p, po, K, i, j, k, rmin, cinv, xi, order);
        gx[tid][0] = xi[0]; gx[tid][1] = xi[1]; gx[tid][2] = xi[2];
        gij[tid][0] = i; gij[tid][1] = j;
        if (blockIdx.x == 0) bin[(base + tid) * K + k] = -r;
    }
    __syncthreads();
    if (c >= ncols) return;
    const double *Ac = A + (size_t)c * a_cs;
    double *oc = Ain + c;
    int ip = -1;
    double ai0 = 0, ai1 = 0, ai2 = 0;
#pragma unroll 4
    for (int e = 0; e < cnt; ++e) {
        const int i = gij[e][0], j = gij[e][1];
        const size_t rj = (size_t)(3 * ((size_t)K * j + k)) * a_rs;
        if (i != ip) {
            const size_t ri = (size_t)(3 * ((size_t)K * i + k)) * a_rs;
            ai0 = Ac[ri]; ai1 = Ac[ri + a_rs]; ai2 = Ac[ri + 2 * a_rs];
            ip = i;
        }
        const double v = -(gx[e][0] * (ai0 - Ac[rj]) + gx[e][1] * (ai1 - Ac[rj + a_rs]) + gx[e][2] * (ai2 - Ac[rj + 2 * a_rs]));
        __builtin_nontemporal_store(v, &oc[(size_t)((base + e) * K + k) * o_rs]);
    }
}

// COLUMN-MAJOR output (o_rs == 1, MATLAB).  Thread = one row (pair, k) with its geometry in registers; block = 256
// consecutive rows x a chunk of RB_CCH columns.  Rows of one pair are consecutive in k, so with a column-major A the
// 3-row blocks a wave reads per column are one contiguous window, and the writes are contiguous along the rows.
#define RB_CCH 32
__global__ __launch_bounds__(256) void add_coll_rows_cm_kernel(int N, int K, const double *__restrict__ p,
                                                               const double *__restrict__ po, double rmin, double cinv,
                                                               const double *__restrict__ A, long a_rs, long a_cs, int ncols,
                                                               double *__restrict__ Ain, long o_cs, double *__restrict__ bin,
                                                               size_t nrows, int order)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const int k = (int)(r % K);
    int i, j;
    unrank_pair((long)(r / K), N, i, j);
    double xi[3];
    const double rr = pair_geom(p, po, K, i, j, k, rmin, cinv, xi, order);
    if (blockIdx.y == 0) bin[r] = -rr;
    const double *Ai = A + (size_t)(3 * ((size_t)K * i + k)) * a_rs, *Aj = A + (size_t)(3 * ((size_t)K * j + k)) * a_rs;
    const int cbeg = blockIdx.y * RB_CCH, cend = min(ncols, cbeg + RB_CCH);
#pragma unroll 4
    for (int c = cbeg; c < cend; ++c) {
        const size_t o = (size_t)c * a_cs;
        const double v = -(xi[0] * (Ai[o] - Aj[o]) + xi[1] * (Ai[o + a_rs] - Aj[o + a_rs]) + xi[2] * (Ai[o + 2 * a_rs] - Aj[o + 2 * a_rs]));
        __builtin_nontemporal_store(v, &Ain[r + (size_t)c * o_cs]);
    }
}

}   // namespace rb
