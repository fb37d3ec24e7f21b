// Microbenchmark: what one SIMD sustains for  { ds_read_b128 of the B operand ; v_mfma_f32_32x32x16_f16 ; V vector
// instructions on the result }  per step, as a function of waves per SIMD, prefetch distance and V — the shape of stage
// C2's filter loop (sc_score.hip).  Prints nominal cycles per step at 2.4 GHz and the real ones from s_memtime / s_memrealtime.
//   hipcc -O3 -w --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit 0: LDS reads on; bit 1: MFMA on; bit 2: epilogue of NV vector instructions on; PF = prefetch distance (1 or 2)
template <int MODE, int NV, int PF>
__global__ __launch_bounds__(256) void k(float* out, int steps, uint64_t* clk) {
  __shared__ uint4 Bt[2048];  // 32 KiB
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) Bt[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  const half8* Bc = reinterpret_cast<const half8*>(Bt) + lane;
  half8 A; for (int i = 0; i < 8; i++) A[i] = (_Float16)(0.001f * (lane + i));
  f32x16 C; for (int i = 0; i < 16; i++) C[i] = (float)i;
  half8 b0 = Bc[0], b1 = Bc[64];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  uint32_t sr[4] = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 2
  for (int g = 0; g < steps; g++) {
    f32x16 D;
    if constexpr ((MODE & 2) != 0) D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b0, C, 0, 0, 0);
    else { D = C; for (int i = 0; i < 16; i++) asm volatile("" : "+v"(D[i])); asm volatile("" ::"v"(b0)); }
    if constexpr ((MODE & 1) != 0) {
      if constexpr (PF == 1) b0 = Bc[64 * ((g + 1) & 31)];
      else { b0 = b1; b1 = Bc[64 * ((g + 2) & 31)]; }
    }
    if constexpr ((MODE & 4) != 0) {
      // NV vector instructions in 4 independent chains on the MFMA's result (3 fma per chain first, like the filter)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float v = __builtin_fmaf(D[4 * j + 2], D[4 * j + 2], __builtin_fmaf(D[4 * j + 1], D[4 * j + 1], __builtin_fmaf(D[4 * j], D[4 * j], -3.0f)));
#pragma unroll
        for (int e = 0; e < (NV - 16) / 4; e++) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        sr[j] = __builtin_amdgcn_alignbit(sr[j], __float_as_uint(v), 31);
      }
    } else {
      sr[0] ^= __float_as_uint(D[0]) ^ __float_as_uint(D[5]);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + (float)(sr[0] ^ sr[1] ^ sr[2] ^ sr[3]);
}

// k2.  KIND 0: the vector instructions do NOT touch the MFMA's result (which chains into the next MFMA's C);
// KIND 1: software pipeline — the vector work of step g - 1 issues after step g's MFMA (two accumulator sets);
// KIND 2: specialised waves — even waves of a workgroup issue only MFMAs, odd waves only vector instructions.
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k2(float* out, int steps, uint64_t* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  half8 A, b; for (int i = 0; i < 8; i++) { A[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.01f * i); }
  f32x16 C; for (int i = 0; i < 16; i++) C[i] = (float)i;
  f32x16 X = C;  // registers the "independent" vector work lives in
  uint32_t sr[4] = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  auto valu = [&](const f32x16& D) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v = __builtin_fmaf(D[4 * j + 2], D[4 * j + 2], __builtin_fmaf(D[4 * j + 1], D[4 * j + 1], __builtin_fmaf(D[4 * j], D[4 * j], -3.0f)));
#pragma unroll
      for (int e = 0; e < (NV - 16) / 4; e++) v = __builtin_fmaf(v, 1.0001f, 0.5f);
      sr[j] = __builtin_amdgcn_alignbit(sr[j], __float_as_uint(v), 31);
    }
  };
  if constexpr (KIND == 0) {
    f32x16 D = C;
    for (int g = 0; g < steps; g++) {
      D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, D, 0, 0, 0);
      valu(X);
#pragma unroll
      for (int j = 0; j < 4; j++) X[4 * j] = __uint_as_float(sr[j] | 0x3f000000u);  // (keeps the chains data dependent on themselves only)
    }
    sr[0] ^= __float_as_uint(D[0]);
  } else if constexpr (KIND == 1) {
    f32x16 D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0), D1;
#pragma unroll 1
    for (int g = 0; g < steps; g += 2) {
      D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
      valu(D0);
      asm volatile("" ::"v"(D0[3]), "v"(D0[7]), "v"(D0[11]), "v"(D0[15]));
      D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
      valu(D1);
      asm volatile("" ::"v"(D1[3]), "v"(D1[7]), "v"(D1[11]), "v"(D1[15]));
    }
    sr[0] ^= __float_as_uint(D0[0]);
  } else {
    if (wave & 1) {
      for (int g = 0; g < 2 * steps; g++) {
        valu(X);
#pragma unroll
        for (int j = 0; j < 4; j++) X[4 * j] = __uint_as_float(sr[j] | 0x3f000000u);
      }
    } else {
      f32x16 D = C;
      for (int g = 0; g < 2 * steps; g++) D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, D, 0, 0, 0);
      sr[0] ^= __float_as_uint(D[0]);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(sr[0] ^ sr[1] ^ sr[2] ^ sr[3]);
}

// k3: the shape of a "Gram-form" filter step: 3 chained MFMAs (K = 48) on three 16-byte LDS reads give 16 squared
// residuals per lane; epilogue = 16 sign-bit shifts + a min tree + one compare (about 25 vector instructions per 1024 tests).
template <int WPB>
__global__ __launch_bounds__(256, WPB) void k3(float* out, int steps, uint64_t* clk) {
  __shared__ uint4 Bt[3 * 512];  // 24 KiB
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 3 * 512; i += 256) Bt[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  const half8* Bc = reinterpret_cast<const half8*>(Bt) + lane;
  half8 A0, A1, A2; for (int i = 0; i < 8; i++) { A0[i] = (_Float16)(0.001f * (lane + i)); A1[i] = (_Float16)(0.002f * i); A2[i] = (_Float16)(0.003f * (lane - i)); }
  f32x16 C; for (int i = 0; i < 16; i++) C[i] = (float)i - 8.f;
  uint32_t sr[16]; for (int i = 0; i < 16; i++) sr[i] = 0;
  uint32_t hits = 0;
  half8 b0 = Bc[0], b1 = Bc[64], b2 = Bc[128];
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int g = 0; g < steps; g++) {
    f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
    b0 = Bc[192 * ((g + 1) & 7)];
    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, b1, D, 0, 0, 0);
    b1 = Bc[192 * ((g + 1) & 7) + 64];
    D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b2, D, 0, 0, 0);
    b2 = Bc[192 * ((g + 1) & 7) + 128];
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      sr[i] = __builtin_amdgcn_alignbit(sr[i], __float_as_uint(D[i]), 31);
      mn = min(mn, __float_as_uint(D[i]));
    }
    if (__builtin_expect(__ballot(mn < 12345u) != 0, 0)) hits++;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  uint32_t x = hits; for (int i = 0; i < 16; i++) x ^= sr[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)x;
}
template <int WPB>
static void run3(const char* name, int wps, float* d, uint64_t* dclk) {
  const int steps = 2000, blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k3<WPB>), dim3(blocks), dim3(256), 0, 0, d, 100, dclk);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k3<WPB>), dim3(blocks), dim3(256), 0, 0, d, steps, dclk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
  const double per_simd = (double)steps * wps;
  printf("%-34s waves/SIMD=%d  %8.3f ms  %6.1f nominal cyc per 1024 tests  clock %.2f GHz  %6.1f real\n", name, wps, ms,
         ms * 1e-3 * 2.4e9 / per_simd, ghz, ms * 1e-3 * ghz * 1e9 / per_simd);
}

template <int KIND, int NV>
static void run2(const char* name, int wps, float* d, uint64_t* dclk) {
  const int steps = 4000, blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<KIND, NV>), dim3(blocks), dim3(256), 0, 0, d, 100, dclk);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k2<KIND, NV>), dim3(blocks), dim3(256), 0, 0, d, steps, dclk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
  const double per_simd = (double)steps * wps;
  printf("%-34s waves/SIMD=%d  %8.3f ms  %6.1f nominal cyc/step  clock %.2f GHz  %6.1f real cyc/step\n", name, wps, ms,
         ms * 1e-3 * 2.4e9 / per_simd, ghz, ms * 1e-3 * ghz * 1e9 / per_simd);
}

template <int MODE, int NV, int PF>
static void run(const char* name, int wps, float* d, uint64_t* dclk) {
  const int steps = 4000, blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV, PF>), dim3(blocks), dim3(256), 0, 0, d, 100, dclk);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV, PF>), dim3(blocks), dim3(256), 0, 0, d, steps, dclk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)clk[0] / (double)clk[1] * 0.1;  // s_memrealtime ticks at 100 MHz
  const double per_simd = (double)steps * wps;
  printf("%-34s waves/SIMD=%d  %8.3f ms  %6.1f nominal cyc/step  clock %.2f GHz  %6.1f real cyc/step\n", name, wps, ms,
         ms * 1e-3 * 2.4e9 / per_simd, ghz, ms * 1e-3 * ghz * 1e9 / per_simd);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  uint64_t* dclk; hipMalloc(&dclk, 16);
  run3<2>("gram step (3 mfma + 3 lds + ~25 valu)", 2, d, dclk);
  run3<3>("gram step (3 mfma + 3 lds + ~25 valu)", 3, d, dclk);
  run3<4>("gram step (3 mfma + 3 lds + ~25 valu)", 4, d, dclk);
  run3<5>("gram step (3 mfma + 3 lds + ~25 valu)", 5, d, dclk);
  for (int w : {8}) {
    run2<0, 16>("mfma + 16 INDEPENDENT valu", w, d, dclk);
    run2<0, 24>("mfma + 24 INDEPENDENT valu", w, d, dclk);
    run2<1, 16>("pipelined mfma + 16 valu", w, d, dclk);
    run2<1, 24>("pipelined mfma + 24 valu", w, d, dclk);
    run2<2, 16>("specialised waves, 16 valu", w, d, dclk);
    run2<2, 24>("specialised waves, 24 valu", w, d, dclk);
  }
  for (int w : {6}) {
    run<2, 16, 1>("mfma only", w, d, dclk);
    run<3, 16, 1>("mfma + lds (pf 1)", w, d, dclk);
    run<3, 16, 2>("mfma + lds (pf 2)", w, d, dclk);
    run<4, 16, 1>("16 valu only", w, d, dclk);
    run<4, 24, 1>("24 valu only", w, d, dclk);
    run<6, 16, 1>("mfma + 16 valu", w, d, dclk);
    run<6, 24, 1>("mfma + 24 valu", w, d, dclk);
    run<7, 16, 1>("mfma + lds + 16 valu (pf 1)", w, d, dclk);
    run<7, 24, 1>("mfma + lds + 24 valu (pf 1)", w, d, dclk);
    run<7, 24, 2>("mfma + lds + 24 valu (pf 2)", w, d, dclk);
    run<5, 24, 1>("lds + 24 valu", w, d, dclk);
  }
  return 0;
}
