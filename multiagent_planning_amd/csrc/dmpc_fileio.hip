// dmpc_fileio.hip -- the reference's on-disk result formats, host code only.
//
//   dmpc_trajectories2file  <-  DMPC::trajectories2file, dmpc/cpp/dmpc.cpp:2088-2126 (read by
//                               dmpc/cpp_results/read_result.m:4-44)
//   dmpc_test2file          <-  test2file, dmpc/cpp/cluster_test.cpp:9-33 (read by dmpc/cpp_results/cluster_test.m)
//
// The reference streams Eigen matrices with `file << M`: default IOFormat = stream precision (6 significant digits,
// %g style), one space between coefficients, every coefficient right-aligned to the widest coefficient of THAT
// matrix.  The writers below reproduce that byte for byte so the reference's MATLAB readers (dlmread) and any
// diff-based tooling see the same files.  Included into dmpc_api.hip (single translation unit).
#include <cstdio>
#include <string>
#include <vector>

namespace fio {

// element (i,j) of a rows x cols matrix at M[i*rs + j*cs]
static void eigen_stream(std::string &out, const double *M, int rows, int cols, long rs, long cs)
{
    char buf[64];
    size_t width = 0;
    for (int j = 0; j < cols; ++j)          // Eigen scans column-major; only the maximum matters
        for (int i = 0; i < rows; ++i) {
            const int n = snprintf(buf, sizeof buf, "%.6g", M[i * rs + j * cs]);
            if ((size_t)n > width) width = (size_t)n;
        }
    for (int i = 0; i < rows; ++i) {
        if (i) out += '\n';
        for (int j = 0; j < cols; ++j) {
            if (j) out += ' ';
            const int n = snprintf(buf, sizeof buf, "%.6g", M[i * rs + j * cs]);
            out.append(width - (size_t)n, ' ');
            out.append(buf, (size_t)n);
        }
    }
}

static void num(std::string &out, double v)
{
    char buf[64];
    out.append(buf, (size_t)snprintf(buf, sizeof buf, "%.6g", v));
}

}   // namespace fio

extern "C" int dmpc_trajectories2file(const char *path, int N, int N_cmd, int T, double h_scaled, const double *pmin,
                                      const double *pmax, const double *po, const double *pf, const double *pos,
                                      const double *vel, const double *acc)
{
    if (!path || N < 1 || N_cmd < 1 || T < 1 || !pmin || !pmax || !po || !pf || !pos || !vel || !acc) {
        g_err = "dmpc_trajectories2file: bad arguments";
        return -1;
    }
    std::string s;
    s.reserve((size_t)N_cmd * T * 3 * 3 * 12 + 4096);
    // file << N << " " << N_cmd << " " << _h_scaled << " " << _pmin.transpose() << " " << _pmax.transpose() << endl;
    s += std::to_string(N); s += ' '; s += std::to_string(N_cmd); s += ' '; fio::num(s, h_scaled); s += ' ';
    fio::eigen_stream(s, pmin, 1, 3, 3, 1); s += ' ';
    fio::eigen_stream(s, pmax, 1, 3, 3, 1); s += '\n';
    fio::eigen_stream(s, po, 3, N, 1, 3); s += '\n';          // _po is 3 x N; po here is [N][3]
    fio::eigen_stream(s, pf, 3, N_cmd, 1, 3); s += '\n';
    const double *blocks[3] = {pos, vel, acc};                 // [N_cmd][T][3] each == MATLAB pk(3,T,N_cmd)
    for (int b = 0; b < 3; ++b)
        for (int i = 0; i < N_cmd; ++i) {
            fio::eigen_stream(s, blocks[b] + (size_t)i * T * 3, 3, T, 1, 3);
            s += '\n';
        }
    FILE *f = fopen(path, "wb");
    if (!f) { g_err = std::string("dmpc_trajectories2file: cannot open ") + path; return -1; }
    const bool ok = fwrite(s.data(), 1, s.size(), f) == s.size();
    if (fclose(f) != 0 || !ok) { g_err = std::string("dmpc_trajectories2file: write failed: ") + path; return -1; }
    return 0;
}

extern "C" int dmpc_test2file(const char *path, int n_cluster, int n_vehicles, int n_trials, const double *cluster_size,
                              const double *num_vehicles, const double *times)
{
    if (!path || n_cluster < 1 || n_vehicles < 1 || n_trials < 1 || !cluster_size || !num_vehicles || !times) {
        g_err = "dmpc_test2file: bad arguments";
        return -1;
    }
    std::string s;
    s += std::to_string(n_cluster); s += ' '; s += std::to_string(n_vehicles); s += ' '; s += std::to_string(n_trials); s += '\n';
    fio::eigen_stream(s, cluster_size, 1, n_cluster, n_cluster, 1); s += ' ';
    fio::eigen_stream(s, num_vehicles, 1, n_vehicles, n_vehicles, 1); s += '\n';
    for (int i = 0; i < n_cluster; ++i) {                      // time_vec.at(i): n_vehicles x n_trials, row-major here
        fio::eigen_stream(s, times + (size_t)i * n_vehicles * n_trials, n_vehicles, n_trials, n_trials, 1);
        s += '\n';
    }
    FILE *f = fopen(path, "wb");
    if (!f) { g_err = std::string("dmpc_test2file: cannot open ") + path; return -1; }
    const bool ok = fwrite(s.data(), 1, s.size(), f) == s.size();
    if (fclose(f) != 0 || !ok) { g_err = std::string("dmpc_test2file: write failed: ") + path; return -1; }
    return 0;
}
