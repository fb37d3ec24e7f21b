// saccot_mex.cpp — MATLAB mex gateway for libsaccot.so.
//
// NEVER RUN UNDER MATLAB: MATLAB / Octave / mex are not installed in the build image, and the upstream repository ships no
// MATLAB code to plug it into (/root/reference/README.md:1-2 is the whole tree).  What IS tested: it compiles with
// -Wall -Wextra -Werror against a declarations-only mex.h in the CPU suite (tests/test_abi.py), and its mexFunction is run
// on the GPU under a stand-in runtime — column-major arrays, parameter struct, logical mask — and compared with the CPU
// restatement (tests/test_gpu_cabi_example.py, tests/mex_stub/).  It is the binding BASELINE.json's north star describes
// ("MATLAB/C++ mex calling HIP through a thin C-ABI layer").
//   build:  mex saccot_mex.cpp -I../include -L../sac-cot_amd -lsaccot
//   use:    [R, t, inl] = saccot_mex(single(src), single(tgt), struct('sigma',0.1,'t_cmp',0.9,'tau',0.1,'min_len',0.1,'T',50000));
//           src, tgt: N x 3 single (column-major = SC_SOA); R 3x3, t 3x1, inl N x 1 logical;  q ~ R p + t
//           optional fields: 'refine' (SC_FLAG_REFINE), 'score_mode' (0 count, 1 truncated MSE, 2 truncated MAE),
//           'devices' = [0 1 2 3]: several GPUs through sc_create_multi / sc_register_multi (RCCL inside the library;
//           one device = plain sc_register).  The device list is fixed by the FIRST call of a MATLAB session.
#include <cstring>

#include "mex.h"
#include "saccot.h"
static sc_multi* g_multi = nullptr;  // one handle for every device count (n_dev == 1 is exactly sc_register)
static void at_exit() { sc_destroy_multi(g_multi); g_multi = nullptr; }

static float fieldf(const mxArray* s, const char* name, float dflt) {
  const mxArray* f = mxGetField(s, 0, name);
  return f ? (float)mxGetScalar(f) : dflt;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 2 || !mxIsSingle(prhs[0]) || !mxIsSingle(prhs[1]) || mxGetN(prhs[0]) != 3 || mxGetN(prhs[1]) != 3 ||
      mxGetM(prhs[0]) != mxGetM(prhs[1]))
    mexErrMsgIdAndTxt("saccot:args", "src and tgt must be N x 3 single");
  if (!g_multi) {
    int devs[64] = {0}, nd = 1;
    const mxArray* dv = nrhs > 2 ? mxGetField(prhs[2], 0, "devices") : nullptr;
    if (dv) {
      nd = (int)mxGetNumberOfElements(dv);
      if (nd < 1 || nd > 64) mexErrMsgIdAndTxt("saccot:args", "devices: 1 .. 64 device ids");
      const double* d = mxGetPr(dv);
      for (int k = 0; k < nd; k++) devs[k] = (int)d[k];
    }
    const int rc = sc_create_multi(devs, nd, &g_multi);
    if (rc != SC_OK) mexErrMsgIdAndTxt("saccot:gpu", "sc_create_multi: %s", sc_strerror(rc));
    mexAtExit(at_exit);
  }
  sc_params p; sc_default_params(&p);
  p.layout = SC_SOA;                                   // MATLAB N x 3 is three planes of N
  if (nrhs > 2) {
    p.sigma = fieldf(prhs[2], "sigma", p.sigma);   p.t_cmp = fieldf(prhs[2], "t_cmp", p.t_cmp);
    p.tau = fieldf(prhs[2], "tau", p.tau);         p.min_len = fieldf(prhs[2], "min_len", p.min_len);
    p.max_triangles = (uint32_t)fieldf(prhs[2], "T", (float)p.max_triangles);
    p.score_mode = (int32_t)fieldf(prhs[2], "score_mode", 0.f);
    if (fieldf(prhs[2], "refine", 0.f) != 0.f) p.flags |= SC_FLAG_REFINE;
  }
  const int64_t n = (int64_t)mxGetM(prhs[0]);
  float R[9], t[3];
  mxArray* inl = mxCreateLogicalMatrix(n, 1);          // mxLogical is 1 byte: written in place
  int rc = sc_register_multi(g_multi, (const float*)mxGetData(prhs[0]), (const float*)mxGetData(prhs[1]), n, &p, R, t,
                             (uint8_t*)mxGetLogicals(inl), nullptr);
  if (rc != SC_OK && rc != SC_ENOHYP) mexErrMsgIdAndTxt("saccot:run", "%s: %s", sc_strerror(rc), sc_multi_last_error(g_multi));
  plhs[0] = mxCreateNumericMatrix(3, 3, mxSINGLE_CLASS, mxREAL);
  float* Rm = (float*)mxGetData(plhs[0]);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rm[c * 3 + r] = R[r * 3 + c];   // row-major -> column-major
  if (nlhs > 1) { plhs[1] = mxCreateNumericMatrix(3, 1, mxSINGLE_CLASS, mxREAL); memcpy(mxGetData(plhs[1]), t, 12); }
  if (nlhs > 2) plhs[2] = inl; else mxDestroyArray(inl);
}
