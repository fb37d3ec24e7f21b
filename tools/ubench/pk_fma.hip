// Microbenchmark: issue cost of v_pk_fma_f32 (two f32 fmas per lane) against v_fma_f32 on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/pk_fma.hip -o gpurun_out/pk_fma && gpurun_out/pk_fma
// Every wave runs ITER trips of 16 independent instructions of one kind; 4 waves per SIMD... (waves per SIMD is an argument).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f2{(float)threadIdx.x + i, (float)i};
  f2 a = {a0, a0 + 1.0f}, b = {b0, b0 * 0.5f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) {  // 2 x v_fma_f32
        asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a.x), "v"(b.x));
      } else if (KIND == 1) {  // 1 x v_pk_fma_f32 (same number of fmas as KIND 0)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else if (KIND == 2) {  // broadcast of the low half of source 0 through op_sel
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b));
      } else if (KIND == 3) {  // v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      } else if (KIND == 4) {  // v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, float* d) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 8 * (KIND == 0 ? 2 : 1) * waves_per_simd;
  const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles per instruction (at 2.4 GHz)  %.1f Tfma/s\n", name, waves_per_simd, ms, cyc,
         (double)iters * 8 * 2 * 64 * 4 * 256 * waves_per_simd / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
  for (int w : {1, 2, 4, 8}) {
    run<0>("2 x v_fma_f32", w, d);
    run<1>("v_pk_fma_f32", w, d);
    run<2>("v_pk_fma_f32 op_sel_hi", w, d);
    run<3>("v_pk_add_f32", w, d);
    run<4>("v_pk_mul_f32", w, d);
  }
  return 0;
}
