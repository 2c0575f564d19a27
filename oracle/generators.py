"""CPU ORACLE for the device start/goal generators (TEST INFRASTRUCTURE ONLY).

Literal restatement of randomTest.m:1-60 and randomExchange.m:1-57 (rejection sampling with MAX_ITER = 200000 and
whole-set restart; the draw-without-replacement permutation) driven by the SAME counter-based stream as
multiagent_planning_amd/csrc/dmpc_generators.hip (splitmix64 of seed, scene, set, draw index), so the device output must
match bit for bit.  MATLAB's own `rand` stream is not reproducible; what is pinned to the reference here is the
algorithm (the .m text), not its random numbers.
"""
import math

import numpy as np

M64 = (1 << 64) - 1
MAX_ITER = 200000


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


class Stream:
    def __init__(self, seed, sid):
        self.base = splitmix64((seed ^ ((sid * 0xD1342543DE82EF95) & M64)) & M64)
        self.ctr = 0

    def draw(self):
        h = splitmix64((self.base + self.ctr) & M64)
        self.ctr += 1
        return float(h >> 11) * (1.0 / 9007199254740992.0)


def _separated(st, N, pmin, pmax, rmin, cinv):
    e = [pmax[d] - pmin[d] for d in range(3)]
    while True:                                                  # randomTest.m:7
        pts = [[pmin[d] + e[d] * st.draw() for d in range(3)]]
        ok = True
        for n in range(1, N):
            tries, placed = 0, False
            while not placed and tries <= MAX_ITER:              # :13
                c = [pmin[d] + e[d] * st.draw() for d in range(3)]
                good = True
                for q in pts:
                    dx, dy, dz = q[0] - c[0], q[1] - c[1], (q[2] - c[2]) * cinv
                    if not math.sqrt(dx * dx + dy * dy + dz * dz) > rmin:   # `if (dist > rmin)`: all elements
                        good = False
                        break
                if good:
                    pts.append(c)
                    placed = True
                tries += 1
            if not placed:                                       # :23-25
                ok = False
                break
        if ok:
            return np.array(pts)


def random_test(S, N, pmin, pmax, rmin, c, seed):
    po, pf = np.zeros((S, N, 3)), np.zeros((S, N, 3))
    for s in range(S):
        po[s] = _separated(Stream(seed, 2 * s), N, pmin, pmax, rmin, 1.0 / c)
        pf[s] = _separated(Stream(seed, 2 * s + 1), N, pmin, pmax, rmin, 1.0 / c)
    return po, pf


def random_exchange(S, N, pmin, pmax, rmin, seed):
    po, pf = np.zeros((S, N, 3)), np.zeros((S, N, 3))
    for s in range(S):
        st = Stream(seed, 2 * s)
        po[s] = _separated(st, N, pmin, pmax, rmin, 1.0)
        array = list(range(N))                                   # randomExchange.m:31-52 (0-based)
        perm = [0] * N
        for i in range(N):
            aux = [x for x in array if x != i]
            if i == N - 1:
                perm[i] = array[0]
            elif i == N - 2 and aux[-1] == N - 1:
                perm[i] = N - 1
            else:
                rng = N - 1 - i                                  # randi([1 N-i]) with the .m's 1-based i
                j = min(int(st.draw() * rng), rng - 1)
                perm[i] = aux[j]
            array.remove(perm[i])
        pf[s] = po[s][perm]
    return po, pf
