// harness.cpp -- calls the MEX gateway's mexFunction() the way MATLAB would, with arguments read from a request file
// (TEST INFRASTRUCTURE, with the mock MEX runtime of this directory).
//   harness <request.bin> <reply.bin>
// request: int32 cmdlen, cmd bytes | int32 K, variant, order | double h, rmin, c, alim, Q1, S1, term, pmin[3], pmax[3], tol |
//          int32 nargs, per argument: int32 ndim, int64 dims[ndim], column-major doubles | int32 nlhs
// reply:   int32 rc (0 ok, 1 = mexErrMsgIdAndTxt: int32 len + message) | int32 nout, per output: int32 class (6 double, 12 int32),
//          int32 ndim, int64 dims[], raw column-major data
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

extern "C" int dmpc_debug_emulate_devices(int n);   // libdmpc_hip.so (development entry, not in the public header)

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short request\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: harness request.bin reply.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("request"); return 2; }
    int32_t cmdlen; rd(f, &cmdlen, 4);
    std::string cmd(cmdlen, '\0'); rd(f, &cmd[0], cmdlen);
    int32_t ip[3]; rd(f, ip, 12);
    double dp[14]; rd(f, dp, sizeof(dp));
    mxArray *prm = mockStruct();
    const char *inames[3] = {"K", "variant", "order"};
    for (int i = 0; i < 3; ++i) mockSetField(prm, inames[i], mxCreateDoubleScalar((double)ip[i]));
    const char *dnames[7] = {"h", "rmin", "c", "alim", "Q1", "S1", "term"};
    for (int i = 0; i < 7; ++i) mockSetField(prm, dnames[i], mxCreateDoubleScalar(dp[i]));
    mxArray *pmin = mxCreateDoubleMatrix(1, 3, mxREAL), *pmax = mxCreateDoubleMatrix(1, 3, mxREAL);
    memcpy(mxGetPr(pmin), dp + 7, 24); memcpy(mxGetPr(pmax), dp + 10, 24);
    mockSetField(prm, "pmin", pmin); mockSetField(prm, "pmax", pmax);
    mockSetField(prm, "tol", mxCreateDoubleScalar(dp[13]));
    int32_t nargs; rd(f, &nargs, 4);
    std::vector<const mxArray *> prhs;
    prhs.push_back(mockString(cmd.c_str()));
    prhs.push_back(prm);
    for (int a = 0; a < nargs; ++a) {
        int32_t ndim; rd(f, &ndim, 4);
        std::vector<int64_t> d64(ndim); rd(f, d64.data(), 8 * (size_t)ndim);
        std::vector<mwSize> dims(d64.begin(), d64.end());
        mxArray *arr = mxCreateNumericArray(dims.size(), dims.data(), mxDOUBLE_CLASS, mxREAL);
        rd(f, mxGetPr(arr), 8 * mxGetNumberOfElements(arr));
        prhs.push_back(arr);
    }
    int32_t nlhs; rd(f, &nlhs, 4);
    fclose(f);
    std::vector<mxArray *> plhs(nlhs > 0 ? nlhs : 1, nullptr);
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("reply"); return 2; }
    int32_t rc = 0;
    // tests of the single-process multi-GPU path on a box with ONE GPU: the gateway's DMPC_DEVICE_ALL context runs this many ranks on
    // the current device (development hook of the library, set by the test harness -- the gateway itself knows nothing of it)
    if (const char *emu = getenv("DMPC_TEST_EMULATE_DEVICES")) dmpc_debug_emulate_devices(atoi(emu));
    try {
        mexFunction(nlhs, plhs.data(), (int)prhs.size(), prhs.data());
    } catch (const std::runtime_error &e) {
        rc = 1;
    }
    fwrite(&rc, 4, 1, o);
    if (rc) {
        const char *m = mockLastError();
        int32_t len = (int32_t)strlen(m);
        fwrite(&len, 4, 1, o); fwrite(m, 1, len, o);
    } else {
        int32_t nout = 0;
        for (int i = 0; i < nlhs; ++i) if (plhs[i]) nout = i + 1;
        fwrite(&nout, 4, 1, o);
        for (int i = 0; i < nout; ++i) {
            const mxArray *a = plhs[i];
            const mwSize nd = mxGetNumberOfDimensions(a);
            const mwSize *d = mxGetDimensions(a);
            // the element size tells double (8) from int32 (4): the mock stores numel * esize bytes
            size_t n = mxGetNumberOfElements(a);
            extern size_t mockByteSize(const mxArray *);
            const size_t bytes = mockByteSize(a);
            const int32_t c = (n && bytes / n == 4) ? 12 : 6;
            const int32_t ndi = (int32_t)nd;
            fwrite(&c, 4, 1, o); fwrite(&ndi, 4, 1, o);
            for (mwSize k = 0; k < nd; ++k) { int64_t v = (int64_t)d[k]; fwrite(&v, 8, 1, o); }
            fwrite(mxGetData(a), 1, bytes, o);
        }
    }
    fclose(o);
    mockRunAtExit();   // the gateway's mexAtExit(cleanup): destroys the persistent context
    return 0;
}
