/* mex.h — DECLARATIONS-ONLY stand-in for MATLAB's mex.h / matrix.h, kept under tests/ so that
 * integration/saccot_mex.cpp meets a compiler in the CPU suite (tests/test_abi.py): header drift between the gateway and
 * include/saccot.h then breaks a test.  It declares the part of the documented MATLAB C Matrix / MEX API the gateway
 * uses, with the documented signatures, and nothing else.  It is NOT MATLAB: the real mex.h is what a maintainer builds
 * against.  tests/mex_stub/mex_host.cpp implements these functions over a plain struct, so that the GPU suite can also
 * RUN the gateway's mexFunction from a C++ program (a stand-in runtime: column-major arrays, struct fields, logicals). */
#ifndef SACCOT_TEST_MEX_STUB_H
#define SACCOT_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS,
               mxDOUBLE_CLASS, mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS,
               mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX } mxComplexity;

bool mxIsSingle(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double* mxGetPr(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
mxLogical* mxGetLogicals(const mxArray* pa);
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateLogicalMatrix(mwSize m, mwSize n);
void mxDestroyArray(mxArray* pa);
int mexAtExit(void (*exit_fcn)(void));
void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...);

/* the gateway's entry point */
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
