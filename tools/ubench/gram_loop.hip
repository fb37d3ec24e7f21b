// Microbenchmark: the loop skeleton of score_gram_kernel (sc_score.hip) with its parts switched on one by one, to see which
// part keeps the matrix pipe from overlapping the vector work.  512-thread workgroups (8 waves x 32 hypotheses), two per CU,
// units of 256 correspondences (2 x 16 KiB in LDS, staged by LDS-DMA), 8 steps per unit, per step two ds_read_b128 + three
// chained v_mfma_f32_32x32x16_f16 + an epilogue.
//   hipcc -O3 -w --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/gram_loop.hip -o /tmp/gram_loop && /tmp/gram_loop
// Flags (template F):  1 barrier per unit   2 LDS-DMA staging   4 16 sign shifts   8 min tree + compare + branch
//                      16 software pipeline by hand (two accumulator sets: chain of step g + 1 issued before the vector work of g)
//                      32 waves 4..7 start half a unit late   64 three INDEPENDENT accumulators instead of a chain
//                      128 s_setprio 1 around the vector work
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int UNIT = 256, TQ = 4, WAVES = 8;

template <int F>
__global__ __launch_bounds__(64 * WAVES, 2) void k(const uint4* __restrict__ tile, uint32_t units, float* out, uint64_t* clk, uint32_t thr) {
  __shared__ uint4 Bt[2][UNIT * TQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto stage = [&](uint32_t u, int buf) {
#pragma unroll
    for (int i = 0; i < UNIT * TQ / (64 * WAVES); i++) {
      const uint4* gsrc = tile + (size_t)u * (UNIT * TQ) + 64 * WAVES * i + tid;
      const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&Bt[buf][64 * WAVES * i + wave * 64]));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  half8 A0, A2;
  for (int i = 0; i < 8; i++) { A0[i] = (_Float16)(0.001f * (lane + i)); A2[i] = (_Float16)(0.003f * (lane - i)); }
  f32x16 C; for (int i = 0; i < 16; i++) C[i] = (float)i + 100.f;
  uint32_t sr[16]; for (int i = 0; i < 16; i++) sr[i] = 0;
  uint32_t hits = 0;
  if ((F & 2) == 0) {  // no staging: fill the buffers once
    for (int i = tid; i < 2 * UNIT * TQ; i += 64 * WAVES) (&Bt[0][0])[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
  } else {
    stage(0, 0);
  }
  auto vec = [&](const f32x16& D) {
    if constexpr ((F & 128) != 0) __builtin_amdgcn_s_setprio(1);
    if constexpr ((F & 4) != 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) sr[i] = __builtin_amdgcn_alignbit(sr[i], __float_as_uint(D[i]), 31);
    } else {
      sr[0] ^= __float_as_uint(D[0]) ^ __float_as_uint(D[5]) ^ __float_as_uint(D[10]) ^ __float_as_uint(D[15]);
    }
    if constexpr ((F & 8) != 0) {
      uint32_t gm[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        gm[j] = min(min(min(__float_as_uint(D[4 * j]), __float_as_uint(D[4 * j + 1])), __float_as_uint(D[4 * j + 2])), __float_as_uint(D[4 * j + 3]));
      const uint32_t mn = min(min(min(gm[0], gm[1]), gm[2]), gm[3]);
      if (__builtin_expect(__ballot(mn < thr) != 0, 0)) hits += gm[1];
    }
    if constexpr ((F & 128) != 0) __builtin_amdgcn_s_setprio(0);
  };
  if constexpr ((F & 32) != 0) {
    if (wave >= 4) __builtin_amdgcn_s_sleep(8);  // ~ half a unit
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t u = 0; u < units; u++) {
    const int buf = (int)(u & 1u);
    if constexpr ((F & 2) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr ((F & 1) != 0) __syncthreads();
    if constexpr ((F & 2) != 0) { if (u + 1 < units) stage((u + 1) % 19, buf ^ 1); }
    const half8* Bc = reinterpret_cast<const half8*>(Bt[buf]) + lane;
    half8 b0 = Bc[0], b1 = Bc[64];
    if constexpr ((F & 16) != 0) {
      f32x16 Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
      Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b1, Da, 0, 0, 0);
      Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, Da, 0, 0, 0);
      b0 = Bc[32 * TQ]; b1 = Bc[32 * TQ + 64];
#pragma unroll
      for (int g = 1; g < UNIT / 32; g += 2) {
        f32x16 Db = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
        Db = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b1, Db, 0, 0, 0);
        Db = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, Db, 0, 0, 0);
        if (g + 1 < UNIT / 32) { b0 = Bc[32 * TQ * (g + 1)]; b1 = Bc[32 * TQ * (g + 1) + 64]; }
        vec(Da);
        if (g + 1 < UNIT / 32) {
          Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
          Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b1, Da, 0, 0, 0);
          Da = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, Da, 0, 0, 0);
          if (g + 2 < UNIT / 32) { b0 = Bc[32 * TQ * (g + 2)]; b1 = Bc[32 * TQ * (g + 2) + 64]; }
        }
        vec(Db);
      }
    } else {
#pragma unroll 1
      for (int g = 0; g < UNIT / 32; g++) {
        f32x16 D;
        if constexpr ((F & 64) != 0) {
          f32x16 D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
          f32x16 D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b1, C, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, C, 0, 0, 0);
          sr[1] ^= __float_as_uint(D1[3]) ^ __float_as_uint(D2[7]);
        } else {
          D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b1, D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, D, 0, 0, 0);
        }
        if (g + 1 < UNIT / 32) { b0 = Bc[32 * TQ * (g + 1)]; b1 = Bc[32 * TQ * (g + 1) + 64]; }
        vec(D);
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  uint32_t x = hits; for (int i = 0; i < 16; i++) x ^= sr[i];
  out[blockIdx.x * 64 * WAVES + tid] = (float)x;
}

template <int F>
static void run(const char* name, const uint4* tile, float* d, uint64_t* dclk) {
  const uint32_t units = 400;
  const int blocks = 512;  // two per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(64 * WAVES), 0, 0, tile, 20u, d, dclk, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(64 * WAVES), 0, 0, tile, units, d, dclk, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t clk[2]; hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
  const double steps_per_simd = (double)units * 8 * 4;  // 4 waves per SIMD
  printf("%-58s %8.3f ms  clock %.2f GHz  %6.1f cycles per step and SIMD (matrix pipe alone: 96)\n", name, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / steps_per_simd);
}

int main() {
  uint4* tile; hipMalloc(&tile, 20 * UNIT * TQ * 16); hipMemset(tile, 0x3c, 20 * UNIT * TQ * 16);
  float* d; hipMalloc(&d, 512 * 512 * 4);
  uint64_t* dclk; hipMalloc(&dclk, 16);
  run<0>("chain only", tile, d, dclk);
  run<64>("three independent MFMAs", tile, d, dclk);
  run<1>("chain + barrier", tile, d, dclk);
  run<3>("chain + barrier + DMA", tile, d, dclk);
  run<4>("chain + 16 shifts", tile, d, dclk);
  run<64 + 4>("independent + 16 shifts", tile, d, dclk);
  run<12>("chain + 16 shifts + min tree", tile, d, dclk);
  run<64 + 12>("independent + 16 shifts + min tree", tile, d, dclk);
  run<15>("chain + all", tile, d, dclk);
  run<15 + 32>("chain + all, waves 4-7 late", tile, d, dclk);
  run<15 + 128>("chain + all, setprio around the vector work", tile, d, dclk);
  run<16 + 4>("hand pipeline + 16 shifts", tile, d, dclk);
  run<16 + 12>("hand pipeline + 16 shifts + min tree", tile, d, dclk);
  run<16 + 15>("hand pipeline + all", tile, d, dclk);
  run<16 + 15 + 32>("hand pipeline + all, waves 4-7 late", tile, d, dclk);
  return 0;
}
