// dmpc_mex.cpp -- MEX gateway binding libdmpc_hip.so (include/dmpc_hip.h) into MATLAB.
//
// Build on a MATLAB host:   mex -I../../include dmpc_mex.cpp -L.. -ldmpc_hip
// (MATLAB's mex.h is not in the build container; there the gateway is compiled and EXECUTED against the mock MEX runtime
// of tests/mock_mex/ -- tests/test_mex_gateway.py.)
//
// One gateway, dispatched on a command string, keeps ONE persistent context per parameter set:
//   [p,v,a,status,info] = dmpc_mex('solve_one', params, l, n, po, vo, ao, pf)
//   [P,V,A,status,info] = dmpc_mex('step_batch', params, l, x_p, x_v, x_a, pf)
//   [r_factor,h_scaled,violation,totdist,traj_time,p] = dmpc_mex('postcheck', params, pk, vk, ak, pf, vmax, amax, Ts)
//   [Lambda,Av,A0,Delta] = dmpc_mex('model_matrices', params)
//   [Ain,bin,dist] = dmpc_mex('coll_rows', params, l, sel0, k_cmp0, k_blk0, p, a0, rmin, c, A)     (0-based indices)
//   [Ain,bin]      = dmpc_mex('add_coll_constr', params, p, po, rmin, c, A)
//   Aaug           = dmpc_mex('posvel_matrix', params)                                  getPosVelMat.m (12 x 3K)
//   [p,v,a]        = dmpc_mex('init_batch', params, po, pf)                             initDMPC.m for N agents (3 x K x N each)
//   [p,v]          = dmpc_mex('prop_state', params, A_p, A_v, A_initp|[], po, vo, a)    propStatedmpc.m / dec-iSCP propState.m
//   [xi,rhs,dist,kc,viol_k,coll] = dmpc_mex('rows_one', params, l, n, po, vo)           CheckCollSoftDMPC + CollConstr*DMPC rows
//   Ain            = dmpc_mex('rows_dense', params, xi, kc, A)                          -diff_mat*A of structured rows
//   [pk,vk,ak,K_T_used,scene_status] = dmpc_mex('transition', params, po, pf, K_T_max, error_tol)   the whole k-loop on the GPU
//   inbounds       = dmpc_mex('is_inbounds', params, p, pmin, pmax)                    is_inbounds.m
//   pass           = dmpc_mex('reached_goal', params, p, pf, error_tol)                ReachedGoal.m
//   tol            = dmpc_mex('max_deviation', params, p, prev_p)                      maxDeviation.m (p, prev_p: 3 x K)
//   [po,pf]        = dmpc_mex('random_test', params, N, pmin, pmax, rmin, c, seed)      randomTest.m
//   [po,pf]        = dmpc_mex('random_exchange', params, N, pmin, pmax, rmin, seed)     randomExchange.m
// `params` is a struct with the fields of dmpc_params (variant as the DMPC_VAR_* integer).
// The signature-preserving wrappers (solveSoftDMPCbound.m, ...) in this directory call 'solve_one'
// and convert status bits into the reference's [] + flag conventions.
#include "mex.h"

#include <cstring>
#include <string>

#include "dmpc_hip.h"

// A few persistent contexts keyed by their parameters (most recently used first): scripts of the reference alternate helpers that
// carry their own constants (initDMPC, CheckCollSoftDMPC, propStatedmpc ...) with the solver inside one loop -- one context re-tuned on
// every call would rebuild and upload its tables twice per agent and MPC step.
struct CtxSlot { dmpc_params prm; dmpc_ctx *ctx; };
static CtxSlot g_slots[4] = {};
static int g_nslots = 0;
static bool g_registered = false;

static void cleanup()
{
    for (int i = 0; i < g_nslots; ++i) if (g_slots[i].ctx) dmpc_destroy(g_slots[i].ctx);
    g_nslots = 0;
}

static double field(const mxArray *s, const char *name)
{
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("dmpc:params", "missing field %s", name);
    return mxGetScalar(f);
}

static dmpc_params read_params(const mxArray *s)
{
    if (!mxIsStruct(s)) mexErrMsgIdAndTxt("dmpc:params", "params must be a struct");
    dmpc_params p{};   // value-initialised: the struct is compared bytewise (context cache)
    p.K = (int32_t)field(s, "K"); p.variant = (int32_t)field(s, "variant");
    p.order = (int32_t)field(s, "order"); p.max_tries = 0;
    p.h = field(s, "h"); p.rmin = field(s, "rmin"); p.c = field(s, "c"); p.alim = field(s, "alim");
    p.Q1 = field(s, "Q1"); p.S1 = field(s, "S1"); p.term = field(s, "term");
    const double *pmin = mxGetPr(mxGetField(s, 0, "pmin")), *pmax = mxGetPr(mxGetField(s, 0, "pmax"));
    for (int d = 0; d < 3; ++d) { p.pmin[d] = pmin[d]; p.pmax[d] = pmax[d]; }
    // optional: weights of the collision-free cost cases (0 = the reference's HEAD constants)
    const mxArray *qf = mxGetField(s, 0, "Qfar"), *qn = mxGetField(s, 0, "Qnear"), *sf = mxGetField(s, 0, "Sfree");
    p.Qfar = qf ? mxGetScalar(qf) : 0.0; p.Qnear = qn ? mxGetScalar(qn) : 0.0; p.Sfree = sf ? mxGetScalar(sf) : 0.0;
    // optional: `tol` of solveDMPC.m:1 (DMPC_VAR_SCP)
    const mxArray *tl = mxGetField(s, 0, "tol");
    p.tol = tl ? mxGetScalar(tl) : 0.0;
    return p;
}

static dmpc_ctx *context(const dmpc_params &p)
{
    for (int i = 0; i < g_nslots; ++i)
        if (std::memcmp(&p, &g_slots[i].prm, sizeof(p)) == 0) {
            const CtxSlot hit = g_slots[i];
            for (int j = i; j > 0; --j) g_slots[j] = g_slots[j - 1];
            g_slots[0] = hit;
            return hit.ctx;
        }
    // every visible GPU (DMPC_DEVICE_ALL): on a multi-GPU node the agents of a scene are sharded over the GPUs inside the library,
    // like the thread clusters of DMPC::solveParallelDMPCv2; with one GPU this is a plain context
    dmpc_ctx *c = nullptr;
    if (g_nslots == 4) {   // re-tune the least recently used context
        c = g_slots[3].ctx;
        if (dmpc_set_params(c, &p)) mexErrMsgIdAndTxt("dmpc:params", "%s", dmpc_last_error(c));
        g_nslots = 3;
    } else {
        // (the library found at run time must speak the header this gateway was compiled with: the special device values changed once)
        if (dmpc_abi_version() != DMPC_ABI_VERSION) mexErrMsgIdAndTxt("dmpc:abi", "libdmpc_hip.so has ABI revision %d, this gateway was compiled for %d", dmpc_abi_version(), DMPC_ABI_VERSION);
        c = dmpc_create(&p, DMPC_DEVICE_ALL, DMPC_PREC_F64);
        if (!c) mexErrMsgIdAndTxt("dmpc:create", "%s", dmpc_last_error(nullptr));
    }
    for (int j = g_nslots; j > 0; --j) g_slots[j] = g_slots[j - 1];
    g_slots[0].prm = p; g_slots[0].ctx = c;
    g_nslots++;
    if (!g_registered) { mexAtExit(cleanup); mexLock(); g_registered = true; }
    return c;
}

static void need(bool ok, const char *what) { if (!ok) mexErrMsgIdAndTxt("dmpc:shape", "%s", what); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    need(nrhs >= 2 && mxIsChar(prhs[0]), "usage: dmpc_mex(cmd, params, ...)");
    char cmd[32];
    mxGetString(prhs[0], cmd, sizeof(cmd));
    dmpc_params p = read_params(prhs[1]);
    const int n3 = 3 * p.K;
    if (!std::strcmp(cmd, "model_matrices")) {
        plhs[0] = mxCreateDoubleMatrix(n3, n3, mxREAL);
        mxArray *Av = mxCreateDoubleMatrix(n3, n3, mxREAL), *A0 = mxCreateDoubleMatrix(6, n3, mxREAL),
                *Dl = mxCreateDoubleMatrix(n3, n3, mxREAL);
        // the C ABI is row-major: the returned column-major MATLAB arrays hold the TRANSPOSES, which the
        // .m wrappers undo (Lambda, A_v, Delta) / A0 is created 6 x 3K and transposed there
        dmpc_model_matrices(&p, mxGetPr(plhs[0]), mxGetPr(Av), mxGetPr(A0), mxGetPr(Dl));
        if (nlhs > 1) plhs[1] = Av; if (nlhs > 2) plhs[2] = A0; if (nlhs > 3) plhs[3] = Dl;
        return;
    }
    if (!std::strcmp(cmd, "posvel_matrix")) {   // getPosVelMat.m:24 (host computation)
        mxArray *t = mxCreateDoubleMatrix(n3, 12, mxREAL);   // row-major 12 x 3K == column-major 3K x 12
        dmpc_posvel_matrix(p.h, p.K, mxGetPr(t));
        plhs[0] = mxCreateDoubleMatrix(12, n3, mxREAL);
        for (int i = 0; i < 12; ++i) for (int j = 0; j < n3; ++j) mxGetPr(plhs[0])[i + 12 * j] = mxGetPr(t)[(size_t)i * n3 + j];
        return;
    }
    dmpc_ctx *ctx = context(p);
    if (!std::strcmp(cmd, "solve_one")) {
        need(nrhs == 8, "solve_one: (cmd, params, l, n, po, vo, ao, pf)");
        const mwSize *dl = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && dl[0] == 3 && (int)dl[1] == p.K, "l must be 3 x K x N");
        const int N = (int)dl[2], n = (int)mxGetScalar(prhs[3]);
        for (int i = 4; i < 8; ++i) need(mxGetNumberOfElements(prhs[i]) == 3, "po, vo, ao, pf must have 3 elements");
        plhs[0] = mxCreateDoubleMatrix(3, p.K, mxREAL);   // column-major 3 x K == stacked [x1 y1 z1 x2 ...]
        mxArray *v = mxCreateDoubleMatrix(3, p.K, mxREAL), *a = mxCreateDoubleMatrix(3, p.K, mxREAL);
        mxArray *st = mxCreateNumericMatrix(1, 1, mxINT32_CLASS, mxREAL), *inf = mxCreateNumericMatrix(1, 8, mxINT32_CLASS, mxREAL);
        // MATLAB l(3,K,N) column-major IS the [N][3K] row-major table: passed without copying
        if (dmpc_solve_one(ctx, N, n - 1, mxGetPr(prhs[2]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6]), mxGetPr(prhs[7]),
                           mxGetPr(plhs[0]), mxGetPr(v), mxGetPr(a), (int32_t *)mxGetData(st), (int32_t *)mxGetData(inf)))
            mexErrMsgIdAndTxt("dmpc:solve", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = v; if (nlhs > 2) plhs[2] = a; if (nlhs > 3) plhs[3] = st; if (nlhs > 4) plhs[4] = inf;
        return;
    }
    if (!std::strcmp(cmd, "step_batch")) {
        need(nrhs == 7, "step_batch: (cmd, params, l, x_p, x_v, x_a, pf)");
        const mwSize *dl = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && dl[0] == 3 && (int)dl[1] == p.K, "l must be 3 x K x N");
        const int N = (int)dl[2];
        for (int i = 3; i < 7; ++i) need(mxGetNumberOfElements(prhs[i]) == (size_t)3 * N, "states must be 3 x N");
        const mwSize d3[3] = {3, (mwSize)p.K, (mwSize)N};
        plhs[0] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        mxArray *v = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL), *a = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        mxArray *st = mxCreateNumericMatrix(1, N, mxINT32_CLASS, mxREAL), *inf = mxCreateNumericMatrix(8, N, mxINT32_CLASS, mxREAL);
        if (dmpc_step_batch(ctx, 1, N, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6]),
                            mxGetPr(plhs[0]), mxGetPr(v), mxGetPr(a), (int32_t *)mxGetData(st), (int32_t *)mxGetData(inf)))
            mexErrMsgIdAndTxt("dmpc:step", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = v; if (nlhs > 2) plhs[2] = a; if (nlhs > 3) plhs[3] = st; if (nlhs > 4) plhs[4] = inf;
        return;
    }
    if (!std::strcmp(cmd, "postcheck")) {   // failure_rate.m:136-195 for one trial
        need(nrhs == 9, "postcheck: (cmd, params, pk, vk, ak, pf, vmax, amax, Ts)");
        const mwSize *dh = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && dh[0] == 3, "pk must be 3 x KT x N");
        const int KT = (int)dh[1], N = (int)dh[2];
        for (int i = 3; i < 5; ++i) need(mxGetNumberOfElements(prhs[i]) == (size_t)3 * KT * N, "vk, ak must match pk");
        need(mxGetNumberOfElements(prhs[5]) == (size_t)3 * N, "pf must be 1 x 3 x N");
        const double vmax = mxGetScalar(prhs[6]), amax = mxGetScalar(prhs[7]), Ts = mxGetScalar(prhs[8]);
        double rf = 0, hs = 0, tot = 0, tt = 0;
        int32_t ns = 0, viol = 0, kt = KT;
        // MATLAB pk(3,KT,N) column-major IS the [N][KT][3] history layout
        if (dmpc_postcheck(ctx, 1, N, KT, &kt, nullptr, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), vmax,
                           amax, Ts, &rf, &hs, &ns, nullptr, &viol, &tot, &tt, nullptr, 0))
            mexErrMsgIdAndTxt("dmpc:postcheck", "%s", dmpc_last_error(ctx));
        plhs[0] = mxCreateDoubleScalar(rf);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(hs);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar((double)viol);
        if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(tot);
        if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(tt);
        if (nlhs > 5) {   // the interpolated positions p(3, length(t), N), second pass now that length(t) is known
            const mwSize d3[3] = {3, (mwSize)ns, (mwSize)N};
            plhs[5] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
            if (dmpc_postcheck(ctx, 1, N, KT, &kt, nullptr, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]),
                               vmax, amax, Ts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mxGetPr(plhs[5]), ns))
                mexErrMsgIdAndTxt("dmpc:postcheck", "%s", dmpc_last_error(ctx));
        }
        return;
    }
    if (!std::strcmp(cmd, "coll_rows")) {   // dec-iSCP/CollConstr.m, dmpc/matlab/CollConstr*DMPC.m
        need(nrhs == 11, "coll_rows: (cmd, params, l, sel0, k_cmp0, k_blk0, p, a0, rmin, c, A)");
        const mwSize *dl = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) >= 2 && dl[0] == 3, "l must be 3 x K x N_obs");
        const int K = (int)dl[1], N_obs = mxGetNumberOfDimensions(prhs[2]) == 3 ? (int)dl[2] : 1;
        const int n_sel = (int)mxGetNumberOfElements(prhs[3]);
        std::string selbuf((size_t)n_sel * sizeof(int32_t) + 1, '\0');
        int32_t *sel = (int32_t *)&selbuf[0];
        for (int i = 0; i < n_sel; ++i) sel[i] = (int32_t)mxGetPr(prhs[3])[i];
        need(mxGetNumberOfElements(prhs[6]) == 3 && mxGetNumberOfElements(prhs[7]) == 3, "p, a0 must have 3 elements");
        const int a_rows = (int)mxGetM(prhs[10]), ncols = (int)mxGetN(prhs[10]);
        plhs[0] = mxCreateDoubleMatrix(n_sel, ncols, mxREAL);
        mxArray *b = mxCreateDoubleMatrix(n_sel, 1, mxREAL), *d = mxCreateDoubleMatrix(n_sel, 1, mxREAL);
        // column-major operands bind through the strides: A(i,j) at i + j*a_rows, Ain(r,c) at r + c*n_sel
        if (dmpc_coll_rows(ctx, K, N_obs, n_sel, sel, mxGetPr(prhs[2]), (int)mxGetScalar(prhs[4]), (int)mxGetScalar(prhs[5]),
                           mxGetPr(prhs[6]), mxGetPr(prhs[7]), mxGetScalar(prhs[8]), mxGetScalar(prhs[9]), mxGetPr(prhs[10]), a_rows,
                           ncols, 1, a_rows, mxGetPr(plhs[0]), 1, n_sel > 0 ? n_sel : 1, mxGetPr(b), mxGetPr(d)))
            mexErrMsgIdAndTxt("dmpc:coll_rows", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = b; if (nlhs > 2) plhs[2] = d;
        return;
    }
    if (!std::strcmp(cmd, "add_coll_constr")) {   // cup-SCP/AddCollConstr.m
        need(nrhs == 7, "add_coll_constr: (cmd, params, p, po, rmin, c, A)");
        const mwSize *dp_ = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && dp_[0] == 3, "p must be 3 x K x N");
        const int K = (int)dp_[1], N = (int)dp_[2];
        need(mxGetNumberOfElements(prhs[3]) == (size_t)3 * N, "po must be 1 x 3 x N");
        need((int)mxGetM(prhs[6]) == 3 * K * N, "A must have 3*K*N rows");
        const int ncols = (int)mxGetN(prhs[6]);
        const mwSize nrows = (mwSize)K * N * (N - 1) / 2;
        plhs[0] = mxCreateDoubleMatrix(nrows, ncols, mxREAL);
        mxArray *b = mxCreateDoubleMatrix(nrows, 1, mxREAL);
        if (dmpc_add_coll_constr(ctx, K, N, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetScalar(prhs[4]), mxGetScalar(prhs[5]),
                                 mxGetPr(prhs[6]), ncols, 1, 3 * K * N, mxGetPr(plhs[0]), 1, (int64_t)nrows, mxGetPr(b)))
            mexErrMsgIdAndTxt("dmpc:add_coll_constr", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = b;
        return;
    }
    if (!std::strcmp(cmd, "init_batch")) {   // initDMPC.m:1-13 for every agent of one scene
        need(nrhs == 4, "init_batch: (cmd, params, po, pf)");
        const int N = (int)(mxGetNumberOfElements(prhs[2]) / 3);
        need(N >= 1 && mxGetNumberOfElements(prhs[2]) == (size_t)3 * N && mxGetNumberOfElements(prhs[3]) == (size_t)3 * N, "po, pf must be 3 x N");
        const mwSize d3[3] = {3, (mwSize)p.K, (mwSize)N};
        plhs[0] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        mxArray *v = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL), *a = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        if (dmpc_init_batch(ctx, 1, N, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(plhs[0]), mxGetPr(v), mxGetPr(a)))
            mexErrMsgIdAndTxt("dmpc:init", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = v; if (nlhs > 2) plhs[2] = a;
        return;
    }
    if (!std::strcmp(cmd, "prop_state")) {   // propStatedmpc.m:1-8 (A_initp given) / dec-iSCP/propState.m:1-10 (A_initp = [])
        need(nrhs == 8, "prop_state: (cmd, params, A_p, A_v, A_initp, po, vo, a)");
        const int rows = (int)mxGetM(prhs[2]), cols = (int)mxGetN(prhs[2]);
        need((int)mxGetM(prhs[3]) == rows && (int)mxGetN(prhs[3]) == cols, "A_p, A_v must have the same shape");
        need((int)mxGetNumberOfElements(prhs[7]) == cols, "a must have one element per column of A_p");
        const bool with_init = mxGetNumberOfElements(prhs[4]) > 0;
        if (with_init) need((int)mxGetM(prhs[4]) == rows && (int)mxGetN(prhs[4]) == 6, "A_initp must be rows x 6");
        // the C ABI takes row-major matrices: transpose the column-major MATLAB operands
        std::string buf((size_t)(2 * rows * cols + rows * 6) * sizeof(double), '\0');
        double *Ap = (double *)&buf[0], *Av = Ap + (size_t)rows * cols, *A0 = Av + (size_t)rows * cols;
        const double *mp = mxGetPr(prhs[2]), *mv = mxGetPr(prhs[3]);
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j) { Ap[(size_t)i * cols + j] = mp[i + (size_t)j * rows]; Av[(size_t)i * cols + j] = mv[i + (size_t)j * rows]; }
        if (with_init) { const double *m0 = mxGetPr(prhs[4]); for (int i = 0; i < rows; ++i) for (int j = 0; j < 6; ++j) A0[i * 6 + j] = m0[i + (size_t)j * rows]; }
        plhs[0] = mxCreateDoubleMatrix(rows, 1, mxREAL);
        mxArray *v = mxCreateDoubleMatrix(rows, 1, mxREAL);
        const double *po = mxGetPr(prhs[5]), *vo = mxGetPr(prhs[6]);
        // propStatedmpc: v = A_v a + repmat(vo); propState: p = A_p a + repmat(po), v = A_v a
        if (dmpc_prop_state(ctx, rows, cols, Ap, Av, with_init ? A0 : nullptr, with_init ? po : nullptr, with_init ? vo : nullptr,
                            with_init ? nullptr : po, with_init ? vo : nullptr, mxGetPr(prhs[7]), mxGetPr(plhs[0]), mxGetPr(v)))
            mexErrMsgIdAndTxt("dmpc:prop_state", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = v;
        return;
    }
    if (!std::strcmp(cmd, "is_inbounds")) {   // is_inbounds.m:1-6: p is 3 x n (column-major = [n][3] points)
        need(nrhs == 5 && mxGetM(prhs[2]) == 3 && mxGetNumberOfElements(prhs[3]) == 3 && mxGetNumberOfElements(prhs[4]) == 3, "is_inbounds: (cmd, params, p(3 x n), pmin, pmax)");
        int32_t ok = 0;
        if (dmpc_is_inbounds(ctx, (int)mxGetN(prhs[2]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), &ok))
            mexErrMsgIdAndTxt("dmpc:is_inbounds", "%s", dmpc_last_error(ctx));
        plhs[0] = mxCreateDoubleScalar(ok ? 1.0 : 0.0);   // (a double 0/1: `if (...)` and `&&` in the scripts take it like the logical of the .m file)
        return;
    }
    if (!std::strcmp(cmd, "reached_goal")) {   // ReachedGoal.m:1-11 on the positions of ONE time index: p, pf are 3 x N
        need(nrhs == 5 && mxGetM(prhs[2]) == 3 && mxGetNumberOfElements(prhs[3]) == mxGetNumberOfElements(prhs[2]), "reached_goal: (cmd, params, p(3 x N), pf(3 x N), error_tol)");
        int32_t ok = 0;
        if (dmpc_reached_goal(ctx, (int)mxGetN(prhs[2]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetScalar(prhs[4]), &ok))
            mexErrMsgIdAndTxt("dmpc:reached_goal", "%s", dmpc_last_error(ctx));
        plhs[0] = mxCreateDoubleScalar(ok ? 1.0 : 0.0);   // (a double 0/1: `if (...)` and `&&` in the scripts take it like the logical of the .m file)
        return;
    }
    if (!std::strcmp(cmd, "max_deviation")) {   // maxDeviation.m:1-11: p, prev_p are 3 x K (column-major = [K][3])
        need(nrhs == 4 && mxGetM(prhs[2]) == 3 && mxGetNumberOfElements(prhs[3]) == mxGetNumberOfElements(prhs[2]), "max_deviation: (cmd, params, p(3 x K), prev_p(3 x K))");
        double tol = 0.0;
        if (dmpc_max_deviation(ctx, (int)mxGetN(prhs[2]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), &tol))
            mexErrMsgIdAndTxt("dmpc:max_deviation", "%s", dmpc_last_error(ctx));
        plhs[0] = mxCreateDoubleScalar(tol);
        return;
    }
    if (!std::strcmp(cmd, "rows_one")) {   // CheckCollSoftDMPC.m + CollConstr*DMPC.m of the context's variant, structured rows
        need(nrhs == 6, "rows_one: (cmd, params, l, n, po, vo)");
        const mwSize *dl = mxGetDimensions(prhs[2]);
        need(mxGetNumberOfDimensions(prhs[2]) == 3 && dl[0] == 3 && (int)dl[1] == p.K, "l must be 3 x K x N");
        const int N = (int)dl[2], n = (int)mxGetScalar(prhs[3]);
        const int cap = (p.variant == DMPC_VAR_HARD ? p.K : (p.variant == DMPC_VAR_ALL3 ? 3 : 1)) * (N > 1 ? N - 1 : 1);
        std::string buf((size_t)cap * (5 * sizeof(double) + sizeof(int32_t)) + 8, '\0');
        double *xi = (double *)&buf[0], *rhs = xi + (size_t)3 * cap, *sc = rhs + cap;
        int32_t *kc = (int32_t *)(sc + cap), nr = 0, vk = 0, st = 0;
        if (dmpc_rows_one(ctx, N, n - 1, mxGetPr(prhs[2]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), cap, xi, rhs, sc, kc, &nr, &vk, &st))
            mexErrMsgIdAndTxt("dmpc:rows", "%s", dmpc_last_error(ctx));
        const int m = nr < cap ? nr : cap;
        plhs[0] = mxCreateDoubleMatrix(3, m, mxREAL);
        std::memcpy(mxGetPr(plhs[0]), xi, (size_t)3 * m * sizeof(double));
        mxArray *b = mxCreateDoubleMatrix(m, 1, mxREAL), *d = mxCreateDoubleMatrix(m, 1, mxREAL), *k = mxCreateDoubleMatrix(m, 1, mxREAL);
        for (int i = 0; i < m; ++i) { mxGetPr(b)[i] = rhs[i]; mxGetPr(d)[i] = sc[i]; mxGetPr(k)[i] = (double)kc[i]; }
        if (nlhs > 1) plhs[1] = b; if (nlhs > 2) plhs[2] = d; if (nlhs > 3) plhs[3] = k;
        if (nlhs > 4) plhs[4] = mxCreateDoubleScalar((double)vk);
        if (nlhs > 5) plhs[5] = mxCreateDoubleScalar((double)((st & DMPC_ST_COLL) != 0));
        return;
    }
    if (!std::strcmp(cmd, "rows_dense")) {   // Ain_total(idx,:) = -diff_mat*Ain (CollConstrSoftDMPC.m:24-27)
        need(nrhs == 5, "rows_dense: (cmd, params, xi, kc, A)");
        const int nr = (int)mxGetNumberOfElements(prhs[3]);
        need((int)mxGetNumberOfElements(prhs[2]) == 3 * nr, "xi must be 3 x nr");
        const int a_rows = (int)mxGetM(prhs[4]), ncols = (int)mxGetN(prhs[4]);
        std::string kb((size_t)nr * sizeof(int32_t) + 4, '\0');
        int32_t *kc = (int32_t *)&kb[0];
        for (int i = 0; i < nr; ++i) kc[i] = (int32_t)mxGetPr(prhs[3])[i];
        plhs[0] = mxCreateDoubleMatrix(nr, ncols, mxREAL);
        if (nr > 0 && dmpc_rows_dense(ctx, nr, mxGetPr(prhs[2]), kc, mxGetPr(prhs[4]), a_rows, ncols, 1, a_rows, mxGetPr(plhs[0]), 1, nr))
            mexErrMsgIdAndTxt("dmpc:rows_dense", "%s", dmpc_last_error(ctx));
        return;
    }
    if (!std::strcmp(cmd, "transition")) {   // the whole `for k = 1:K_T` loop (dmpc_soft_bound.m:115-148) of one trial
        need(nrhs == 6, "transition: (cmd, params, po, pf, K_T_max, error_tol)");
        const int N = (int)(mxGetNumberOfElements(prhs[2]) / 3), KT = (int)mxGetScalar(prhs[4]);
        need(N >= 1 && mxGetNumberOfElements(prhs[3]) == (size_t)3 * N && KT >= 2, "po, pf must be 3 x N, K_T_max >= 2");
        const mwSize d3[3] = {3, (mwSize)KT, (mwSize)N};
        plhs[0] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        mxArray *v = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL), *a = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
        int32_t used = 0, sst = 0;
        if (dmpc_transition(ctx, 1, N, mxGetPr(prhs[2]), mxGetPr(prhs[3]), KT, mxGetScalar(prhs[5]), mxGetPr(plhs[0]), mxGetPr(v),
                            mxGetPr(a), &used, &sst))
            mexErrMsgIdAndTxt("dmpc:transition", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = v; if (nlhs > 2) plhs[2] = a;
        if (nlhs > 3) plhs[3] = mxCreateDoubleScalar((double)used);
        if (nlhs > 4) plhs[4] = mxCreateDoubleScalar((double)sst);
        return;
    }
    if (!std::strcmp(cmd, "random_test") || !std::strcmp(cmd, "random_exchange")) {   // randomTest.m / randomExchange.m
        const bool ex = cmd[7] == 'e';
        need(nrhs == (ex ? 7 : 8), ex ? "random_exchange: (cmd, params, N, pmin, pmax, rmin, seed)" : "random_test: (cmd, params, N, pmin, pmax, rmin, c, seed)");
        const int N = (int)mxGetScalar(prhs[2]);
        need(N >= 1 && mxGetNumberOfElements(prhs[3]) == 3 && mxGetNumberOfElements(prhs[4]) == 3, "pmin, pmax must have 3 elements");
        plhs[0] = mxCreateDoubleMatrix(3, N, mxREAL);
        mxArray *pf = mxCreateDoubleMatrix(3, N, mxREAL);
        const int rc = ex ? dmpc_random_exchange(ctx, 1, N, mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetScalar(prhs[5]), (uint64_t)mxGetScalar(prhs[6]),
                                                 mxGetPr(plhs[0]), mxGetPr(pf))
                          : dmpc_random_test(ctx, 1, N, mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetScalar(prhs[5]), mxGetScalar(prhs[6]),
                                             (uint64_t)mxGetScalar(prhs[7]), mxGetPr(plhs[0]), mxGetPr(pf));
        if (rc) mexErrMsgIdAndTxt("dmpc:random", "%s", dmpc_last_error(ctx));
        if (nlhs > 1) plhs[1] = pf;
        return;
    }
    mexErrMsgIdAndTxt("dmpc:cmd", "unknown command %s", cmd);
}
