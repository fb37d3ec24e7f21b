// mex_host.cpp — a stand-in MEX runtime for the test suite: implements the functions tests/mex_stub/mex.h declares over
// a plain struct and calls integration/saccot_mex.cpp's mexFunction the way MATLAB would:
//     [R, t, inl] = saccot_mex(single(src), single(tgt), struct('tau', tau, 'sigma', tau, 'min_len', tau, 'T', T))
// on a correspondence file (sac-cot_amd/corrio.py's text format: n lines of px py pz qx qy qz).  Prints R (row-major),
// t and the inlier count; tests/test_gpu_cabi_example.py compares them with the CPU restatement.  Test infrastructure only.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
  mxClassID cls = mxUNKNOWN_CLASS;
  size_t m = 0, n = 0;
  std::vector<unsigned char> data;
  std::map<std::string, mxArray*> fields;  // mxSTRUCT_CLASS (1 x 1)
};

static size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxLOGICAL_CLASS: case mxINT8_CLASS: case mxUINT8_CLASS: return 1;
    default: return 0;
  }
}
static mxArray* make(mxClassID c, size_t m, size_t n) {
  mxArray* a = new mxArray;
  a->cls = c; a->m = m; a->n = n;
  a->data.assign(m * n * elem_size(c), 0);
  return a;
}
static void (*g_at_exit)(void) = nullptr;

extern "C" {
bool mxIsSingle(const mxArray* pa) { return pa && pa->cls == mxSINGLE_CLASS; }
size_t mxGetM(const mxArray* pa) { return pa->m; }
size_t mxGetN(const mxArray* pa) { return pa->n; }
size_t mxGetNumberOfElements(const mxArray* pa) { return pa->m * pa->n; }
void* mxGetData(const mxArray* pa) { return const_cast<unsigned char*>(pa->data.data()); }
double* mxGetPr(const mxArray* pa) { return pa->cls == mxDOUBLE_CLASS ? reinterpret_cast<double*>(const_cast<unsigned char*>(pa->data.data())) : nullptr; }
double mxGetScalar(const mxArray* pa) {
  if (pa->cls == mxDOUBLE_CLASS) return *reinterpret_cast<const double*>(pa->data.data());
  if (pa->cls == mxSINGLE_CLASS) return *reinterpret_cast<const float*>(pa->data.data());
  return 0.0;
}
mxLogical* mxGetLogicals(const mxArray* pa) { return pa->cls == mxLOGICAL_CLASS ? reinterpret_cast<mxLogical*>(const_cast<unsigned char*>(pa->data.data())) : nullptr; }
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname) {
  if (!pa || pa->cls != mxSTRUCT_CLASS || index != 0) return nullptr;
  auto it = pa->fields.find(fieldname);
  return it == pa->fields.end() ? nullptr : it->second;
}
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity) { return make(classid, m, n); }
mxArray* mxCreateLogicalMatrix(mwSize m, mwSize n) { return make(mxLOGICAL_CLASS, m, n); }
void mxDestroyArray(mxArray* pa) {
  if (!pa) return;
  for (auto& kv : pa->fields) mxDestroyArray(kv.second);
  delete pa;
}
int mexAtExit(void (*f)(void)) { g_at_exit = f; return 0; }
void mexErrMsgIdAndTxt(const char* identifier, const char* err_msg, ...) {
  va_list ap; va_start(ap, err_msg);
  fprintf(stderr, "%s: ", identifier);
  vfprintf(stderr, err_msg, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  exit(3);  // MATLAB unwinds to the prompt; the stand-in ends the program
}
}  // extern "C"

static mxArray* scalar(double v) {
  mxArray* a = make(mxDOUBLE_CLASS, 1, 1);
  memcpy(a->data.data(), &v, 8);
  return a;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: mex_host <correspondences.txt> <tau> <T> [refine]\n"); return 2; }
  FILE* f = fopen(argv[1], "r");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<float> rows;
  char line[512];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    float v[6];
    if (sscanf(line, "%f %f %f %f %f %f", v, v + 1, v + 2, v + 3, v + 4, v + 5) == 6) rows.insert(rows.end(), v, v + 6);
  }
  fclose(f);
  const size_t n = rows.size() / 6;
  mxArray* src = make(mxSINGLE_CLASS, n, 3);  // column-major N x 3, as MATLAB holds it
  mxArray* tgt = make(mxSINGLE_CLASS, n, 3);
  float* ps = reinterpret_cast<float*>(src->data.data());
  float* pt = reinterpret_cast<float*>(tgt->data.data());
  for (size_t i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) { ps[c * n + i] = rows[6 * i + c]; pt[c * n + i] = rows[6 * i + 3 + c]; }
  mxArray* opt = new mxArray;
  opt->cls = mxSTRUCT_CLASS; opt->m = opt->n = 1;
  const double tau = atof(argv[2]);
  opt->fields["tau"] = scalar(tau); opt->fields["sigma"] = scalar(tau); opt->fields["min_len"] = scalar(tau);
  opt->fields["t_cmp"] = scalar(0.9); opt->fields["T"] = scalar(atof(argv[3]));
  if (argc > 4) opt->fields["refine"] = scalar(atof(argv[4]));
  const mxArray* prhs[3] = {src, tgt, opt};
  mxArray* plhs[3] = {nullptr, nullptr, nullptr};
  mexFunction(3, plhs, 3, prhs);
  const float* R = reinterpret_cast<const float*>(plhs[0]->data.data());  // 3 x 3 column-major
  printf("R");
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) printf(" %.9g", R[c * 3 + r]);
  const float* t = reinterpret_cast<const float*>(plhs[1]->data.data());
  printf("\nt %.9g %.9g %.9g\n", t[0], t[1], t[2]);
  size_t inl = 0;
  const mxLogical* L = mxGetLogicals(plhs[2]);
  for (size_t i = 0; i < n; i++) inl += L[i] ? 1 : 0;
  printf("n %zu inliers %zu mask", n, inl);
  for (size_t i = 0; i < n; i++) putchar(L[i] ? '1' : '0');
  printf("\n");
  for (int k = 0; k < 3; k++) mxDestroyArray(plhs[k]);
  mxDestroyArray(src); mxDestroyArray(tgt); mxDestroyArray(opt);
  if (g_at_exit) g_at_exit();
  return 0;
}
