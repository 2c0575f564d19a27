// dmpc_generators.hip -- the reference's start/goal generators on the device.
//
//   randomTest.m:1-60      N points uniform in the box, each farther than rmin (ellipsoidal, E1 = diag(1,1,1/c)) from
//                          all earlier ones, by rejection (<= 200000 tries per point, else the whole set restarts);
//                          start set and goal set drawn independently
//   randomExchange.m:1-57  starts the same way (Euclidean), goals = starts permuted so that no agent keeps its own
//
// One wave per (scene, set): the points live in LDS, the 64 lanes test a candidate against the earlier points in
// parallel.  MATLAB's global `rand` stream cannot be reproduced, so the stream is a counter-based one (splitmix64 of
// (seed, scene, set, draw index)); oracle/generators.py restates kernel and stream and must agree bit for bit.
//
// Included into dmpc_api.hip (single translation unit).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gen {

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// draw number `ctr` of stream (seed, sid): uniform in [0, 1) with 53 random bits
__host__ __device__ inline double uniform(uint64_t seed, uint64_t sid, uint64_t ctr)
{
    const uint64_t h = splitmix64(splitmix64(seed ^ (sid * 0xD1342543DE82EF95ull)) + ctr);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

constexpr int MAX_ITER = 200000;

// grid = S * nsets waves; pts_out[nsets][S][N][3]; exchange: set 1 is the permuted copy of set 0 (one wave per scene)
__global__ __launch_bounds__(64) void random_points_kernel(int S, int N, int nsets, int exchange, double pmin0, double pmin1,
                                                           double pmin2, double pmax0, double pmax1, double pmax2, double rmin,
                                                           double cinv, uint64_t seed, double *__restrict__ out)
{
#pragma clang fp contract(off)   // the oracle restatement has no fused multiply-add
    extern __shared__ double pts[];   // [N][3] (+ exchange: N ints of the permutation work array)
    const int lane = threadIdx.x;
    const int scene = blockIdx.x % S, set = blockIdx.x / S;
    const uint64_t sid = (uint64_t)scene * 2 + (uint64_t)set;
    uint64_t ctr = 0;
    const double e0 = pmax0 - pmin0, e1 = pmax1 - pmin1, e2 = pmax2 - pmin2;
    bool pass = false;
    while (!pass) {                                     // randomTest.m:7 -- restart the set from scratch
        if (lane == 0) {
            pts[0] = pmin0 + e0 * uniform(seed, sid, ctr); pts[1] = pmin1 + e1 * uniform(seed, sid, ctr + 1);
            pts[2] = pmin2 + e2 * uniform(seed, sid, ctr + 2);
        }
        ctr += 3;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        pass = true;
        for (int n = 1; n < N && pass; ++n) {
            int tries = 0;
            bool placed = false;
            while (!placed && tries <= MAX_ITER) {      // :13
                const double c0 = pmin0 + e0 * uniform(seed, sid, ctr), c1 = pmin1 + e1 * uniform(seed, sid, ctr + 1),
                             c2 = pmin2 + e2 * uniform(seed, sid, ctr + 2);
                ctr += 3;
                bool close = false;
                for (int j = lane; j < n; j += 64) {
                    const double dx = pts[3 * j] - c0, dy = pts[3 * j + 1] - c1, dz = (pts[3 * j + 2] - c2) * cinv;
                    const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                    close |= !(dist > rmin);            // `if (dist > rmin)` on a vector: all elements
                }
                if (!__any(close)) {
                    if (lane == 0) { pts[3 * n] = c0; pts[3 * n + 1] = c1; pts[3 * n + 2] = c2; }
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                    placed = true;
                }
                ++tries;
            }
            if (!placed) pass = false;                  // :23-25
        }
    }
    double *o = out + ((size_t)set * S + scene) * (size_t)N * 3;
    for (int i = lane; i < 3 * N; i += 64) o[i] = pts[i];
    if (exchange) {
        // randomExchange.m:30-52: perm(i) drawn from the ids still unassigned, never i itself
        int *array = (int *)(pts + 3 * (size_t)N);       // the unassigned ids, in increasing order
        int *perm = array + N;
        if (lane == 0) {
            int len = N;
            for (int i = 0; i < N; ++i) array[i] = i;
            for (int i = 0; i < N; ++i) {
                // array_aux = array without i
                int pos_i = -1;
                for (int t = 0; t < len; ++t) if (array[t] == i) { pos_i = t; break; }
                int pick;
                if (i == N - 1) pick = array[0];                                           // :38-39
                else {
                    const int last_aux = (pos_i == len - 1) ? array[len - 2] : array[len - 1];
                    if (i == N - 2 && last_aux == N - 1) pick = N - 1;                      // :40-42
                    else {
                        // randi([1 N-i]) with the .m's 1-based i: the first N-1-i entries of array_aux (when i itself was
                        // already given away array_aux has one more entry, which the reference never draws)
                        const int range = N - 1 - i;
                        int j = (int)(uniform(seed, sid, ctr) * (double)range);
                        ctr += 1;
                        if (j >= range) j = range - 1;
                        pick = array[(pos_i >= 0 && j >= pos_i) ? j + 1 : j];
                    }
                }
                perm[i] = pick;
                int t = 0;
                while (array[t] != pick) ++t;
                for (; t + 1 < len; ++t) array[t] = array[t + 1];
                --len;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        double *o1 = out + ((size_t)1 * S + scene) * (size_t)N * 3;
        for (int i = lane; i < 3 * N; i += 64) o1[i] = pts[3 * perm[i / 3] + i % 3];
    }
}

}   // namespace gen
