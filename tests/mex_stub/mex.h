/* Declaration-only stand-in for MATLAB's mex.h -- TEST INFRASTRUCTURE, nothing links against it.
 *
 * The build image has no MATLAB, so mex/gsmcal_mex.c can never be built here.  tests/test_abi_cpu.py compiles every
 * gateway target with `gcc -fsyntax-only` against these declarations: a check of syntax and of every call into
 * include/gsmcal.h (argument counts and types), not of MATLAB semantics.  Only the subset of the published MEX C API
 * that the gateway uses is declared -- the interleaved-complex one (-R2018a) by default, the split-complex one
 * (mxGetPr / mxGetPi, every release and still the default of `mex`) with -DGSMCAL_STUB_SPLIT.  Each mode declares ONLY its
 * own accessors, so a gateway path that reaches for the other API's functions does not compile. */
#ifndef GSMCAL_TEST_MEX_STUB_H
#define GSMCAL_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdint.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct { double real, imag; } mxComplexDouble;
typedef uint8_t mxUint8;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef unsigned char mxLogical;
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
int mxIsComplex(const mxArray*);
int mxIsUint8(const mxArray*);
double mxGetScalar(const mxArray*);
#ifdef GSMCAL_STUB_SPLIT
#define MX_HAS_INTERLEAVED_COMPLEX 0
double* mxGetPr(const mxArray*);
double* mxGetPi(const mxArray*);
void* mxGetData(const mxArray*);
#else
#define MX_HAS_INTERLEAVED_COMPLEX 1
double* mxGetDoubles(const mxArray*);
mxComplexDouble* mxGetComplexDoubles(const mxArray*);
mxUint8* mxGetUint8s(const mxArray*);
#endif
int mexPrintf(const char* fmt, ...);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateLogicalScalar(mxLogical v);
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
void mxSetCell(mxArray* cell, mwIndex i, mxArray* v);
void* mxMalloc(size_t);
void* mxCalloc(size_t, size_t);
void mxFree(void*);
int mexAtExit(void (*fn)(void));
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
