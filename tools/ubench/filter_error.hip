// How far is the filter's fp16-split residual from the true one, against the bound eta it assumes?
//   bash tools/ubench/run.sh filter_error
// Same operand construction as score_filter_kernel (sc_score.hip): E' = 1024 s (R p + t - q) through one
// v_mfma_f32_32x32x16_f16 per 8 hypotheses x 32 correspondences, fp16 hi / lo splits, t through C.  For every test the
// kernel compares E' / (1024 s) with the residual evaluated in fp64 from the same fp32 inputs, and with the canonical
// fp32 chain; it reports the largest vector error in units of eta = 2^-16 (2.6 Pmax + Qmax + Tmax) / s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float RS = 1024.0f;
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__global__ __launch_bounds__(64) void k_err(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt, uint32_t ldl,
                                            float s, float pmax, float qmax, double* __restrict__ out /* [0] max err/eta vs fp64, [1] vs fp32 chain, [2] max |canon - fp64| / eta */) {
  const int lane = threadIdx.x, col = lane & 31, hf = lane >> 5;
  const uint32_t h0 = blockIdx.x * 8;
  half8 A;
  {
    const int r = lane & 31, hy = r >> 2, c = r & 3;
    _Float16 rh[3], rl[3];
    for (int kk = 0; kk < 3; kk++) {
      const float v = c < 3 ? Rt[(size_t)(3 * c + kk) * ldl + h0 + hy] * RS : 0.f;
      rh[kk] = (_Float16)v; rl[kk] = (_Float16)(v - (float)rh[kk]);
    }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A = hf ? a1 : a0;
  }
  f32x16 C;
  float tmax = 0.f;
  for (int hh = 0; hh < 8; hh++) for (int c = 0; c < 3; c++) tmax = fmaxf(tmax, fabsf(Rt[(size_t)(9 + c) * ldl + h0 + hh]));
  for (int jj = 0; jj < 4; jj++) {
    const uint32_t h = h0 + 2 * jj + hf;
    C[4 * jj] = Rt[(size_t)9 * ldl + h] * RS * s; C[4 * jj + 1] = Rt[(size_t)10 * ldl + h] * RS * s; C[4 * jj + 2] = Rt[(size_t)11 * ldl + h] * RS * s; C[4 * jj + 3] = 0.f;
  }
  const double eta = (2.6 * pmax * s + qmax * s + tmax * s) / 65536.0 / s;  // in unscaled units
  double w64 = 0, w32 = 0, wc = 0;
  for (int g = 0; g < n; g += 32) {
    const int m = g + col;
    float v[6];
    for (int c = 0; c < 6; c++) v[c] = m < n ? planes[(size_t)c * ld + m] : 0.f;
    _Float16 hi[6], lo[6];
    for (int c = 0; c < 6; c++) { const float X = v[c] * s; hi[c] = (_Float16)X; lo[c] = (_Float16)(X - (float)hi[c]); }
    const half8 f0 = {hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2]};
    const half8 f1 = {hi[2], hi[3], lo[3], hi[4], lo[4], hi[5], lo[5], (_Float16)0.f};
    const half8 b = hf ? f1 : f0;
    const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
    if (m < n)
      for (int jj = 0; jj < 4; jj++) {
        const uint32_t h = h0 + 2 * jj + hf;
        float M[12];
        for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + h];
        double e64[3]; float e32[3];
        for (int c = 0; c < 3; c++) {
          e64[c] = (double)M[3 * c] * v[0] + (double)M[3 * c + 1] * v[1] + (double)M[3 * c + 2] * v[2] + (double)M[9 + c] - (double)v[3 + c];
          e32[c] = M[9 + c] + fma_(M[3 * c + 2], v[2], fma_(M[3 * c + 1], v[1], fma_(M[3 * c], v[0], -v[3 + c])));
        }
        double d64 = 0, d32 = 0, dc = 0;
        for (int c = 0; c < 3; c++) {
          const double ea = (double)D[4 * jj + c] / ((double)RS * s);
          d64 += (ea - e64[c]) * (ea - e64[c]); d32 += (ea - (double)e32[c]) * (ea - (double)e32[c]);
          dc += ((double)e32[c] - e64[c]) * ((double)e32[c] - e64[c]);
        }
        w64 = fmax(w64, sqrt(d64) / eta); w32 = fmax(w32, sqrt(d32) / eta); wc = fmax(wc, sqrt(dc) / eta);
      }
  }
  for (int o = 32; o > 0; o >>= 1) { w64 = fmax(w64, __shfl_xor(w64, o)); w32 = fmax(w32, __shfl_xor(w32, o)); wc = fmax(wc, __shfl_xor(wc, o)); }
  if (lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(out);
    atomicMax(&o[0], (unsigned long long)__double_as_longlong(w64)); atomicMax(&o[1], (unsigned long long)__double_as_longlong(w32));
    atomicMax(&o[2], (unsigned long long)__double_as_longlong(wc));
  }
}

int main() {
  const int n = 4096, ld = 4096; const uint32_t T = 4096;
  printf("largest |filter residual - reference| over %u x %d tests, in units of the bound eta the filter assumes\n", T, n);
  for (int trial = 0; trial < 6; trial++) {
    const float L = (float[]){1.f, 3.f, 50.f, 1000.f, 0.01f, 3.f}[trial];
    const float rot_noise = trial == 5 ? 0.3f : 0.02f;  // trial 5: sloppy "rotations" (entries up to ~1.3)
    std::vector<float> planes(6 * (size_t)ld), Rt(12 * (size_t)T);
    srand(trial + 1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    float pmax = 0, qmax = 0;
    const float ang = 0.7f + trial, cg = cosf(ang), sg = sinf(ang), tg[3] = {0.3f * L, -0.4f * L, 0.45f * L};
    for (int m = 0; m < n; m++) {
      float p[3] = {rnd() * L / 2, rnd() * L / 2, rnd() * L / 2};
      float q[3] = {cg * p[0] - sg * p[1] + tg[0], sg * p[0] + cg * p[1] + tg[1], p[2] + tg[2]};
      if (m % 3) for (int c = 0; c < 3; c++) q[c] = rnd() * L;
      for (int c = 0; c < 3; c++) { planes[c * ld + m] = p[c]; planes[(3 + c) * ld + m] = q[c]; pmax = fmaxf(pmax, fabsf(p[c])); qmax = fmaxf(qmax, fabsf(q[c])); }
    }
    for (uint32_t h = 0; h < T; h++) {
      const float a = ang + 0.05f * rnd(), ch = cosf(a), sh = sinf(a);
      const float R[9] = {ch, -sh, 0, sh, ch, 0, 0, 0, 1};
      for (int c = 0; c < 9; c++) Rt[(size_t)c * T + h] = R[c] + rot_noise * rnd();
      for (int c = 0; c < 3; c++) Rt[(size_t)(9 + c) * T + h] = tg[c] + 0.1f * L * rnd();
    }
    const float mx = fmaxf(pmax, qmax);
    int e; frexpf(mx, &e);  // mx in [2^(e-1), 2^e)
    const float s = ldexpf(1.f, 8 - (e - 1));
    float *d_pl, *d_Rt; double* d_out;
    hipMalloc(&d_pl, planes.size() * 4); hipMalloc(&d_Rt, Rt.size() * 4); hipMalloc(&d_out, 24);
    hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_Rt, Rt.data(), Rt.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_out, 0, 24);
    hipLaunchKernelGGL(k_err, dim3(T / 8), dim3(64), 0, 0, d_pl, n, ld, d_Rt, T, s, pmax, qmax, d_out);
    double o[3]; hipMemcpy(o, d_out, 24, hipMemcpyDeviceToHost);
    printf("extent %8g  scale 2^%-3d  rotation noise %.2f:  vs fp64 %.4f eta   vs the fp32 chain %.4f eta   (fp32 chain vs fp64: %.4f eta)\n", L,
           (int)log2f(s), rot_noise, o[0], o[1], o[2]);
    hipFree(d_pl); hipFree(d_Rt); hipFree(d_out);
  }
  return 0;
}
