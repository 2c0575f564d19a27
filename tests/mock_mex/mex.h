/* mex.h -- MOCK of the subset of MATLAB's MEX API that multiagent_planning_amd/matlab/dmpc_mex.cpp uses (TEST
 * INFRASTRUCTURE: MATLAB and its mex.h do not exist in the build image; SURVEY.md 8b suggests exactly this).  It lets the
 * gateway be COMPILED here, and -- with mock_mex.cpp -- EXECUTED: tests/mock_mex/harness.cpp calls mexFunction() the way
 * MATLAB would (column-major mxArrays, a params struct, a command string).  Semantics follow the documented MATLAB API:
 * column-major data, mxGetM/N = first dimension / product of the rest, mexErrMsgIdAndTxt does not return. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>

typedef size_t mwSize;
typedef struct mxArray_tag mxArray;
typedef enum { mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxCHAR_CLASS = 4, mxSTRUCT_CLASS = 2 } mxClassID;
typedef enum { mxREAL = 0 } mxComplexity;

#ifdef __cplusplus
extern "C" {
#endif
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

bool mxIsStruct(const mxArray *a);
bool mxIsChar(const mxArray *a);
mxArray *mxGetField(const mxArray *a, mwSize index, const char *name);
double mxGetScalar(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
const mwSize *mxGetDimensions(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);   /* does not return (throws in the mock) */
int mexAtExit(void (*fn)(void));
void mexLock(void);
/* mock-only constructors used by the harness */
mxArray *mockString(const char *s);
mxArray *mockStruct(void);
void mockSetField(mxArray *s, const char *name, mxArray *value);
void mockDestroy(mxArray *a);
const char *mockLastError(void);
void mockRunAtExit(void);
#ifdef __cplusplus
}
#endif
#endif
