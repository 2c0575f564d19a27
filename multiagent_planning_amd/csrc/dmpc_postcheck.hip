// dmpc_postcheck.hip -- whole-transition post-checks on the device.
//
// What the reference does after every transition (test/failure_rate.m:136-195, same block in comp_kctr.m,
// comp_hardsoft2.m, dmpc_soft_bound.m:152-190): rescale the MPC solution to the velocity / acceleration limits,
// interpolate at 100 Hz with MATLAB `spline` (not-a-knot cubic), check every agent pair for an ellipsoidal
// collision, and measure path length and trajectory time.  Histories are laid out [S][N][KT_alloc][3] (the
// layout dmpc_transition records), so one (scene, agent, axis) series has stride 3 doubles.
//
// Included into dmpc_api.hip (single translation unit).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pc {

__device__ __forceinline__ double block_min(double v, double *sh)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sh[t] = fmin(sh[t], sh[t + s]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// r_factor = min over agents and knots of min(amax/|a_k|, vmax/|v_k|)   (failure_rate.m:138-145)
__global__ void rfactor_kernel(int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ vk,
                               const double *__restrict__ ak, double vmax, double amax, double *__restrict__ rf)
{
    __shared__ double sh[256];
    const int s = blockIdx.x, KT = kt_used[s];
    double m = INFINITY;
    for (int e = threadIdx.x; e < N * KT; e += blockDim.x) {
        const int i = e / KT, k = e - i * KT;
        const size_t o = (((size_t)s * N + i) * KTa + k) * 3;
        const double an = sqrt(ak[o] * ak[o] + ak[o + 1] * ak[o + 1] + ak[o + 2] * ak[o + 2]);
        const double vn = sqrt(vk[o] * vk[o] + vk[o + 1] * vk[o + 1] + vk[o + 2] * vk[o + 2]);
        m = fmin(m, fmin(amax / an, vmax / vn));
    }
    m = block_min(m, sh);
    if (threadIdx.x == 0) rf[s] = m;
}

// a_k *= r; v_{k+1} = v_k + hs a_k; p_{k+1} = p_k + hs v_k + hs^2/2 a_k   (failure_rate.m:156-162)
// one thread per (scene, agent, axis); writes the rescaled knots y (and v, a) in place
__global__ void rescale_kernel(int S, int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ rf,
                               const double *__restrict__ hs, double *__restrict__ pk, double *__restrict__ vk,
                               double *__restrict__ ak)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)S * N * 3) return;
    const int ax = (int)(g % 3);
    const size_t sa = g / 3;
    const int s = (int)(sa / N);
    const int KT = kt_used[s];
    const double r = rf[s], h = hs[s], h22 = h * h / 2;
    const size_t o = sa * (size_t)KTa * 3 + ax;
    double p = pk[o], v = vk[o];
    for (int k = 0; k + 1 < KT; ++k) {
        const double a = ak[o + (size_t)k * 3] * r;
        ak[o + (size_t)k * 3] = a;
        const double vn = v + h * a;
        p = p + h * v + h22 * a;
        v = vn;
        vk[o + (size_t)(k + 1) * 3] = v;
        pk[o + (size_t)(k + 1) * 3] = p;
    }
}

// Not-a-knot cubic spline on uniform knots (MATLAB spline(tk, y)): second derivatives M_k of one series.
//   M0 - 2 M1 + M2 = 0,  M_{k-1} + 4 M_k + M_{k+1} = 6 (y_{k+1} - 2 y_k + y_{k-1}) / h^2,  same at the far end.
// Eliminating the end rows gives M_1 = d_1/6, M_{n-2} = d_{n-2}/6 and a (1,4,1) system for 2..n-3 (Thomas).
// one thread per (scene, agent, axis); `w` is per-series scratch of the same shape as M
__global__ void spline_kernel(int S, int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ hs,
                              const double *__restrict__ y, double *__restrict__ M, double *__restrict__ w)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)S * N * 3) return;
    const int ax = (int)(g % 3);
    const size_t sa = g / 3;
    const int s = (int)(sa / N);
    const int n = kt_used[s];
    const size_t o = sa * (size_t)KTa * 3 + ax;
#define Y(k) y[o + (size_t)(k) * 3]
#define MM(k) M[o + (size_t)(k) * 3]
#define W(k) w[o + (size_t)(k) * 3]
    if (n < 4) {   // spline() degenerates to the parabola / line through the points: constant second derivative
        const double h = hs[s];
        const double m = (n == 3) ? (Y(2) - 2 * Y(1) + Y(0)) / (h * h) : 0.0;
        for (int k = 0; k < n; ++k) MM(k) = m;
        return;
    }
    const double s6 = 6.0 / (hs[s] * hs[s]);
    const double m1 = (Y(2) - 2 * Y(1) + Y(0)) * s6 / 6.0;
    const double me = (Y(n - 1) - 2 * Y(n - 2) + Y(n - 3)) * s6 / 6.0;
    MM(1) = m1;
    MM(n - 2) = me;
    // forward sweep over 2..n-3
    double cp = 0.0, dp = 0.0;
    for (int k = 2; k <= n - 3; ++k) {
        double d = (Y(k + 1) - 2 * Y(k) + Y(k - 1)) * s6;
        if (k == 2) d -= m1;
        if (k == n - 3) d -= me;
        const double den = 4.0 - ((k == 2) ? 0.0 : cp);
        cp = 1.0 / den;
        dp = (d - ((k == 2) ? 0.0 : dp)) / den;
        W(k) = cp;
        MM(k) = dp;
    }
    for (int k = n - 4; k >= 2; --k) MM(k) = MM(k) - W(k) * MM(k + 1);
    MM(0) = 2 * MM(1) - MM(2);
    MM(n - 1) = 2 * MM(n - 2) - MM(n - 3);
#undef W
}

__device__ __forceinline__ double spline_eval(const double *__restrict__ y, const double *__restrict__ M, size_t o, int n,
                                              double h, double t)
{
    int k = (int)floor(t / h);
    k = k < 0 ? 0 : (k > n - 2 ? n - 2 : k);
    const double u = t - k * h;
    const double y0 = Y(k), y1 = Y(k + 1), m0 = MM(k), m1 = MM(k + 1);
    const double b = (y1 - y0) / h - h * (2 * m0 + m1) / 6.0;
    return y0 + u * (b + u * (m0 / 2 + u * (m1 - m0) / (6.0 * h)));
}
// the same evaluation for the kernels below the #undefs
__device__ __forceinline__ double spline_eval2(const double *__restrict__ y, const double *__restrict__ M, size_t o, int n, double h, double t)
{
    return spline_eval(y, M, o, n, h, t);
}
#undef Y
#undef MM

// squared ellipsoidal distance |E1 (p_i - p_j)|^2 of one pair (failure_rate.m:172): ONE expression with pinned contractions for
// every kernel that evaluates it, so the brute-force and the cell-grid search return the same bits
__device__ __forceinline__ double pair_d2(double xi, double yi, double zi, double xj, double yj, double zj, double cinv)
{
    const double dx = xi - xj, dy = yi - yj, dz = (zi - zj) * cinv;
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

#define PC_SAMPLES_PER_BLOCK 8
// pairwise ellipsoidal distance at every 100 Hz sample (failure_rate.m:165-181): block = (sample group, scene);
// positions of all agents at one sample are staged in LDS, pairs are strided over the threads; the per-scene
// minimum is order independent so an integer atomicMin on the (non-negative) double's bit pattern is exact.
__global__ void pairdist_kernel(int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ hs,
                                const int *__restrict__ ns, double Ts, double cinv, const double *__restrict__ y,
                                const double *__restrict__ M, unsigned long long *__restrict__ mind2,
                                double *__restrict__ p_interp, int ns_alloc)
{
    extern __shared__ double pos[];   // [N][3]
    __shared__ double sh[256];
    const int s = blockIdx.y, n = kt_used[s], nsamp = ns[s];
    const double h = hs[s];
    double m = INFINITY;
    for (int q = 0; q < PC_SAMPLES_PER_BLOCK; ++q) {
        const int smp = blockIdx.x * PC_SAMPLES_PER_BLOCK + q;
        if (smp >= nsamp) break;
        const double t = smp * Ts;
        for (int e = threadIdx.x; e < N * 3; e += blockDim.x) {
            const size_t o = ((size_t)s * N + e / 3) * (size_t)KTa * 3 + e % 3;
            const double v = spline_eval(y, M, o, n, h, t);
            pos[e] = v;
            if (p_interp && smp < ns_alloc) p_interp[(((size_t)s * N + e / 3) * ns_alloc + smp) * 3 + e % 3] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
            const int i = e / N, j = e - i * N;
            if (j <= i) continue;
            m = fmin(m, pair_d2(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], cinv));
        }
        __syncthreads();
    }
    m = block_min(m, sh);
    if (threadIdx.x == 0 && m < INFINITY) atomicMin(&mind2[s], (unsigned long long)__double_as_longlong(m));
}

// ---------------------------------------------------------------------------------------------------------------------
// Large scenes (N > PC_BRUTE_MAX): the all-pairs search of failure_rate.m:170-181 is O(N^2) per 100 Hz sample -- 5e7 pairs x
// 3 000 samples at N = 10^4.  Per (scene, sample) the agents are binned into a uniform grid whose cells are `edge` wide in
// the metric of the check (x, y, z / c): two agents closer than `edge` lie in the same or in adjacent cells, so testing
// the 27-neighbourhood finds EVERY pair with distance < edge, and each of those is evaluated with the exact fp64 expression
// of the brute-force search.  Hence: if the minimum the grid search returns is <= edge it is the scene's exact minimum
// (bit for bit what the brute force returns: a minimum does not depend on the order of its operands); if it is larger or no
// pair was found at all, no pair is closer than edge >= 2 rmin -- the verdict "no violation" is already proven, and the
// scene is searched again by brute force only to report the exact `min_dist`.  Cell indices are clamped to the grid, which
// keeps adjacency (clamping is monotone and non-expansive), so positions outside the workspace cost candidates, not
// correctness.  A batch of SB samples is processed per pass: evaluate + count, exclusive scan, scatter, search.
struct Grid {
    int nx, ny, nz;
    double x0, y0, z0;      // lower corner
    double inv_e, inv_ez;   // 1 / edge (x, y) and 1 / (edge c) (z)
};
#define PC_BRUTE_MAX 256

// thread per (scene, sample of the batch, agent): spline position -> pts, cell -> cell_of, count -> fill
__global__ void grid_eval_kernel(int S, int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ hs,
                                 const int *__restrict__ ns, double Ts, int smp0, int SB, const double *__restrict__ y,
                                 const double *__restrict__ M, Grid g, double *__restrict__ pts, int *__restrict__ cell_of,
                                 int *__restrict__ fill, double *__restrict__ p_interp, int ns_alloc)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)S * SB * N) return;
    const int i = (int)(t % N);
    const int b = (int)((t / N) % SB), s = (int)(t / ((size_t)N * SB));
    const int smp = smp0 + b;
    if (smp >= ns[s]) { cell_of[t] = -1; return; }
    const int n = kt_used[s];
    const double h = hs[s], tt = smp * Ts;
    const size_t o = ((size_t)s * N + i) * (size_t)KTa * 3;
    const double x = spline_eval2(y, M, o, n, h, tt), yv = spline_eval2(y, M, o + 1, n, h, tt), z = spline_eval2(y, M, o + 2, n, h, tt);
    pts[3 * t] = x; pts[3 * t + 1] = yv; pts[3 * t + 2] = z;
    if (p_interp && smp < ns_alloc) {
        double *d = p_interp + (((size_t)s * N + i) * ns_alloc + smp) * 3;
        d[0] = x; d[1] = yv; d[2] = z;
    }
    int ix = (int)floor((x - g.x0) * g.inv_e), iy = (int)floor((yv - g.y0) * g.inv_e), iz = (int)floor((z - g.z0) * g.inv_ez);
    ix = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= g.nz ? g.nz - 1 : iz);
    const int c = ix + g.nx * (iy + g.ny * iz);
    cell_of[t] = c;
    atomicAdd(&fill[((size_t)s * SB + b) * ((size_t)g.nx * g.ny * g.nz) + c], 1);
}
// block per (scene, sample): start[c] = number of agents in cells < c (ncell + 1 entries); the counts go back to zero
__global__ void grid_scan_kernel(int ncell, int *__restrict__ fill, int *__restrict__ start)
{
    __shared__ int part[1024];
    int *f = fill + (size_t)blockIdx.x * ncell, *st = start + (size_t)blockIdx.x * (ncell + 1);
    const int per = (ncell + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < ncell ? lo + per : ncell;
    int a = 0;
    for (int c = lo; c < hi; ++c) a += f[c];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive scan of the partial sums
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - a;
    for (int c = lo; c < hi; ++c) { st[c] = run; run += f[c]; f[c] = 0; }
    if (threadIdx.x == 1023) st[ncell] = part[1023];
}
// thread per (scene, sample, agent): slot of the agent in its cell (the order inside a cell is arbitrary: only a minimum is taken)
__global__ void grid_scatter_kernel(size_t total, int N, int ncell, const int *__restrict__ cell_of, const int *__restrict__ start,
                                    int *__restrict__ fill, int *__restrict__ sorted)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = cell_of[t];
    if (c < 0) return;
    const size_t sb = t / N;
    const int slot = start[sb * (ncell + 1) + c] + atomicAdd(&fill[sb * ncell + c], 1);
    sorted[sb * N + slot] = (int)(t % N);
}
// block = (tile of 256 agents i, sample, scene): every agent j > i of the 27 cells around i's
__global__ void grid_pairs_kernel(int N, int SB, Grid g, double cinv, const double *__restrict__ pts, const int *__restrict__ cell_of,
                                  const int *__restrict__ start, const int *__restrict__ sorted, unsigned long long *__restrict__ mind2)
{
    __shared__ double sh[256];
    const int s = blockIdx.z, b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const size_t sb = (size_t)s * SB + b, t = sb * N + i;
    double m = INFINITY;
    const int c = i < N ? cell_of[t] : -1;
    if (c >= 0) {
        const int ncell = g.nx * g.ny * g.nz;
        const int ix = c % g.nx, iy = (c / g.nx) % g.ny, iz = c / (g.nx * g.ny);
        const double xi = pts[3 * t], yi = pts[3 * t + 1], zi = pts[3 * t + 2];
        const int *st = start + sb * (ncell + 1), *so = sorted + sb * N;
        const double *pp = pts + sb * (size_t)N * 3;
        const int x_lo = ix > 0 ? ix - 1 : 0, x_hi = ix + 1 < g.nx ? ix + 1 : g.nx - 1;
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = iz + dz;
            if (z < 0 || z >= g.nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = iy + dy;
                if (yy < 0 || yy >= g.ny) continue;
                const int base = g.nx * (yy + g.ny * z);
                const int e0 = st[base + x_lo], e1 = st[base + x_hi + 1];   // x is the fastest cell index: three cells = one run
                for (int e = e0; e < e1; ++e) {
                    const int j = so[e];
                    if (j <= i) continue;
                    m = fmin(m, pair_d2(xi, yi, zi, pp[3 * j], pp[3 * j + 1], pp[3 * j + 2], cinv));
                }
            }
        }
    }
    m = block_min(m, sh);
    if (threadIdx.x == 0 && m < INFINITY) atomicMin(&mind2[s], (unsigned long long)__double_as_longlong(m));
}
// brute force over the positions of a batch (fallback of the grid search: scenes without any pair closer than `edge`):
// block = (tile of 256 agents i, sample, scene); the tiles of j >= tile(i) stream through LDS
__global__ void pairs_brute_pts_kernel(int N, int SB, double cinv, const double *__restrict__ pts, const int *__restrict__ cell_of,
                                       const int *__restrict__ scene_on, unsigned long long *__restrict__ mind2)
{
    __shared__ double tile[256 * 3];
    __shared__ double sh[256];
    const int s = blockIdx.z, b = blockIdx.y, it = blockIdx.x;
    if (!scene_on[s]) return;
    const size_t sb = (size_t)s * SB + b;
    if (cell_of[sb * N] < 0) return;   // sample beyond the end of this scene's transition
    const double *pp = pts + sb * (size_t)N * 3;
    const int i = it * 256 + threadIdx.x;
    const bool vi = i < N;
    const double xi = vi ? pp[3 * i] : 0.0, yi = vi ? pp[3 * i + 1] : 0.0, zi = vi ? pp[3 * i + 2] : 0.0;
    double m = INFINITY;
    for (int jt = it; jt * 256 < N; ++jt) {
        const int j0 = jt * 256, cntj = N - j0 < 256 ? N - j0 : 256;
        __syncthreads();
        for (int e = threadIdx.x; e < cntj * 3; e += 256) tile[e] = pp[3 * (size_t)j0 + e];
        __syncthreads();
        if (vi)
            for (int jj = 0; jj < cntj; ++jj) {
                if (j0 + jj <= i) continue;
                m = fmin(m, pair_d2(xi, yi, zi, tile[3 * jj], tile[3 * jj + 1], tile[3 * jj + 2], cinv));
            }
    }
    m = block_min(m, sh);
    if (threadIdx.x == 0 && m < INFINITY) atomicMin(&mind2[s], (unsigned long long)__double_as_longlong(m));
}

// per agent: path length sum |p(t_{s+1}) - p(t_s)| (failure_rate.m:183) and the 1-based index after the last
// sample farther than 5 cm from the goal (failure_rate.m:186-193)
__global__ void path_kernel(int S, int N, int KTa, const int *__restrict__ kt_used, const double *__restrict__ hs,
                            const int *__restrict__ ns, double Ts, const double *__restrict__ y, const double *__restrict__ M,
                            const double *__restrict__ pf, double *__restrict__ dist, int *__restrict__ tidx)
{
    const size_t sa = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sa >= (size_t)S * N) return;
    const int s = (int)(sa / N), n = kt_used[s], nsamp = ns[s];
    const double h = hs[s];
    const size_t o = sa * (size_t)KTa * 3;
    const double gx = pf[sa * 3], gy = pf[sa * 3 + 1], gz = pf[sa * 3 + 2];
    double px = 0, py = 0, pz = 0, acc = 0;
    int last = 0;
    for (int smp = 0; smp < nsamp; ++smp) {
        const double t = smp * Ts;
        const double x = spline_eval(y, M, o, n, h, t), yv = spline_eval(y, M, o + 1, n, h, t),
                     z = spline_eval(y, M, o + 2, n, h, t);
        if (smp) acc += sqrt((x - px) * (x - px) + (yv - py) * (yv - py) + (z - pz) * (z - pz));
        px = x; py = yv; pz = z;
        const double dg = sqrt((x - gx) * (x - gx) + (yv - gy) * (yv - gy) + (z - gz) * (z - gz));
        if (dg >= 0.05) last = smp + 2;
    }
    dist[sa] = acc;
    tidx[sa] = last;
}

// fixed-order per-scene reductions of the per-agent results
__global__ void finish_kernel(int N, const double *__restrict__ dist, const int *__restrict__ tidx, double Ts,
                              double *__restrict__ totdist, double *__restrict__ traj_time)
{
    __shared__ double sd[256];
    __shared__ int si[256];
    const int s = blockIdx.x, t = threadIdx.x;
    double a = 0;
    int m = 0;
    for (int i = t; i < N; i += blockDim.x) { a += dist[(size_t)s * N + i]; m = max(m, tidx[(size_t)s * N + i]); }
    sd[t] = a; si[t] = m;
    __syncthreads();
    for (int k = blockDim.x >> 1; k > 0; k >>= 1) {
        if (t < k) { sd[t] += sd[t + k]; si[t] = max(si[t], si[t + k]); }
        __syncthreads();
    }
    if (t == 0) { totdist[s] = sd[0]; traj_time[s] = si[0] * Ts; }
}

}   // namespace pc
