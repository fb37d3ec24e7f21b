// ubench_valu.hip — what one MI355X SIMD issues per clock for fp32 FMA: plain v_fma_f32 vs v_pk_fma_f32,
// as a function of waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1e-3f + c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fmaf(x[c], a, b);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += x[c];
  if (s == 12345.678f) out[0] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters, float a, float b) {
  float2_ x[CHAINS];
  float2_ va = {a, a}, vb = {b, b};
#pragma unroll
  for (int c = 0; c < CHAINS; c++) x[c] = float2_{threadIdx.x * 1e-3f + c, threadIdx.x * 2e-3f + c};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[c]) : "v"(x[c]), "v"(va), "v"(vb));
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += x[c].x + x[c].y;
  if (s == 12345.678f) out[0] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000; constexpr int CH = 8;
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 wave per SIMD per block
    int blocks = 256 * wps;
    float ms1 = timeit([&] { hipLaunchKernelGGL(k_fma<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
    float ms2 = timeit([&] { hipLaunchKernelGGL(k_pkfma<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
    double fl1 = 2.0 * CH * iters * 256.0 * blocks, fl2 = 2.0 * fl1;
    printf("waves/SIMD %d: v_fma_f32 %.1f TFLOP/s (%.3f ms)   v_pk_fma_f32 %.1f TFLOP/s (%.3f ms)\n", wps,
           fl1 / ms1 / 1e9, ms1, fl2 / ms2 / 1e9, ms2);
  }
  return 0;
}
