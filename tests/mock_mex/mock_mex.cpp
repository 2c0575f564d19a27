// mock_mex.cpp -- minimal runtime behind the mock mex.h (TEST INFRASTRUCTURE).
#include "mex.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    std::vector<mwSize> dims;
    std::vector<unsigned char> data;
    std::map<std::string, mxArray *> fields;
    std::string str;
};
static std::string g_err;
static void (*g_atexit)(void) = nullptr;

static size_t numel(const mxArray *a) { size_t n = 1; for (mwSize d : a->dims) n *= d; return n; }
static mxArray *make(mxClassID cls, const std::vector<mwSize> &dims, size_t esize)
{
    mxArray *a = new mxArray_tag();
    a->cls = cls; a->dims = dims;
    a->data.assign(numel(a) * esize, 0);
    return a;
}
extern "C" {
bool mxIsStruct(const mxArray *a) { return a && a->cls == mxSTRUCT_CLASS; }
bool mxIsChar(const mxArray *a) { return a && a->cls == mxCHAR_CLASS; }
mxArray *mxGetField(const mxArray *a, mwSize, const char *name)
{
    auto it = a->fields.find(name);
    return it == a->fields.end() ? nullptr : it->second;
}
double mxGetScalar(const mxArray *a)
{
    if (a->cls == mxINT32_CLASS) return (double)*(const int32_t *)a->data.data();
    return *(const double *)a->data.data();
}
double *mxGetPr(const mxArray *a) { return (double *)a->data.data(); }
void *mxGetData(const mxArray *a) { return (void *)a->data.data(); }
int mxGetString(const mxArray *a, char *buf, mwSize buflen)
{
    if (a->str.size() + 1 > buflen) return 1;
    std::strcpy(buf, a->str.c_str());
    return 0;
}
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims.data(); }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->dims.size(); }
size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
size_t mxGetM(const mxArray *a) { return a->dims.empty() ? 0 : a->dims[0]; }
size_t mxGetN(const mxArray *a) { size_t n = 1; for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i]; return n; }
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(mxDOUBLE_CLASS, {m, n}, 8); }
mxArray *mxCreateDoubleScalar(double v) { mxArray *a = make(mxDOUBLE_CLASS, {1, 1}, 8); *(double *)a->data.data() = v; return a; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) { return make(cls, {m, n}, cls == mxINT32_CLASS ? 4 : 8); }
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity)
{
    return make(cls, std::vector<mwSize>(dims, dims + ndim), cls == mxINT32_CLASS ? 4 : 8);
}
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = std::string(id) + ": " + buf;
    throw std::runtime_error(g_err);
}
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
void mexLock(void) {}
mxArray *mockString(const char *s) { mxArray *a = new mxArray_tag(); a->cls = mxCHAR_CLASS; a->str = s; a->dims = {1, std::strlen(s)}; return a; }
mxArray *mockStruct(void) { mxArray *a = new mxArray_tag(); a->cls = mxSTRUCT_CLASS; a->dims = {1, 1}; return a; }
void mockSetField(mxArray *s, const char *name, mxArray *value) { s->fields[name] = value; }
void mockDestroy(mxArray *a) { if (!a) return; for (auto &kv : a->fields) mockDestroy(kv.second); delete a; }
const char *mockLastError(void) { return g_err.c_str(); }
void mockRunAtExit(void) { if (g_atexit) g_atexit(); g_atexit = nullptr; }
}
size_t mockByteSize(const mxArray *a) { return a->data.size(); }
