// Microbenchmark for stage C2 (inlier counting): where does the matrix-pipe body lose its time?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 \
//         tools/ubench/score_mfma.hip -o gpurun_out/score_mfma && gpurun_out/score_mfma
// T = 50 176 hypotheses x N = 5000 correspondences (BASELINE configs[2]); every variant must print the same checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PC = 512;  // points per chunk
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------- V: lane = hypothesis, plain VALU (the product's default body) ----------------
__global__ __launch_bounds__(256, 8) void k_valu(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                 uint32_t ldl, float tau2, uint32_t* __restrict__ partial) {
  __shared__ float4 smem[2 * PC];
  float4* pA = smem; float2* pB = reinterpret_cast<float2*>(smem + PC);
  const int m0 = blockIdx.y * PC, cnt = min(PC, n - m0), padded = (cnt + 3) & ~3;
  for (int t = threadIdx.x; t < padded; t += 256) {
    const int m = m0 + t;
    if (t < cnt) { pA[t] = make_float4(planes[m], planes[ld + m], planes[2 * ld + m], planes[3 * ld + m]); pB[t] = make_float2(planes[4 * ld + m], planes[5 * ld + m]); }
    else { pA[t] = make_float4(0, 0, 0, 1e30f); pB[t] = make_float2(1e30f, 1e30f); }
  }
  const uint32_t l = blockIdx.x * 256 + threadIdx.x;
  float M[12];
#pragma unroll
  for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + l];
  __syncthreads();
  uint32_t c0 = 0;
#pragma clang loop unroll_count(4)
  for (int t = 0; t < padded; t++) {
    const float4 a = pA[t]; const float2 b = pB[t];
    const float ex = M[9] + fma_(M[2], a.z, fma_(M[1], a.y, fma_(M[0], a.x, -a.w)));
    const float ey = M[10] + fma_(M[5], a.z, fma_(M[4], a.y, fma_(M[3], a.x, -b.x)));
    const float ez = M[11] + fma_(M[8], a.z, fma_(M[7], a.y, fma_(M[6], a.x, -b.y)));
    const float d2 = fma_(ez, ez, fma_(ey, ey, ex * ex));
    c0 += d2 < tau2 ? 1u : 0u;
  }
  partial[(size_t)blockIdx.y * ldl + l] = c0;
}

// ---------------- M: matrix pipe.  MODE 0: MFMAs only (rate check; wrong counts)  1: + epilogue  ----------------
// QUADS hypothesis quads per wave; 4 waves per workgroup.
template <int QUADS, int MODE>
__global__ __launch_bounds__(256, 8) void k_mfma(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                 uint32_t ldl, float tau2, uint32_t* __restrict__ partial, int rep = 1) {
  __shared__ float4 smem[2 * PC];
  float4* P4 = smem; float4* Qn = smem + PC;
  const int m0 = blockIdx.y * PC, cntp = min(PC, n - m0), padded = (cntp + 15) & ~15;
  for (int t = threadIdx.x; t < padded; t += 256) {
    const int m = m0 + t;
    if (t < cntp) { P4[t] = make_float4(planes[m], planes[ld + m], planes[2 * ld + m], 1.0f); Qn[t] = make_float4(-planes[3 * ld + m], -planes[4 * ld + m], -planes[5 * ld + m], 0.f); }
    else { P4[t] = make_float4(0, 0, 0, 1.0f); Qn[t] = make_float4(-1e30f, -1e30f, -1e30f, 0.f); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t hyp0 = (blockIdx.x * 4 + wave) * (4 * QUADS);
  const int a_hq = (lane & 15) >> 2, a_c = lane & 3, a_k = lane >> 4;
  float a[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const uint32_t h = hyp0 + 4 * q + a_hq;
    a[q] = a_c < 3 ? Rt[(size_t)(a_k < 3 ? 3 * a_c + a_k : 9 + a_c) * ldl + h] : 0.0f;
  }
  __syncthreads();
  uint32_t cnt[QUADS];
  f32x4 acc[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; q++) { cnt[q] = 0; acc[q] = f32x4{0, 0, 0, 0}; }
  const float* P4f = reinterpret_cast<const float*>(P4);
  const int col = lane & 15, kb = lane >> 4;
  for (int r = 0; r < rep; r++)
  for (int g = 0; g < padded; g += 16) {
    const float b = P4f[(g + col) * 4 + kb];
    const float4 cq = Qn[g + col];
    const f32x4 c = {cq.x, cq.y, cq.z, cq.w};
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < QUADS; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b, acc[q], 0, 0, 0);
    } else {
      f32x4 d[QUADS];
#pragma unroll
      for (int q = 0; q < QUADS; q++) d[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b, c, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < QUADS; q++) {
        const float d2 = fma_(d[q][2], d[q][2], fma_(d[q][1], d[q][1], d[q][0] * d[q][0]));
        cnt[q] += (d2 < tau2) ? 1u : 0u;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    uint32_t c = MODE == 0 ? (uint32_t)(acc[q][0] + acc[q][1] + acc[q][2]) : cnt[q];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 16);
    const uint32_t h = hyp0 + 4 * q + (lane >> 4);
    if (col == 0) partial[(size_t)blockIdx.y * ldl + h] = c;
  }
}

// ---------------- S: matrix pipe + counting on the scalar unit ----------------
// v_cmp leaves a 64-bit lane mask per MFMA; masks are summed in bit-sliced (vertical) counters held in SGPR pairs:
// plane b of counter q holds bit b of every lane's count.  One 3:2 compressor per mask: 5 scalar ops, no VALU at all.
template <int QUADS>
__global__ __launch_bounds__(256, 8) void k_mfma_salu(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                      uint32_t ldl, float tau2, uint32_t* __restrict__ partial) {
  __shared__ float4 smem[2 * PC];
  float4* P4 = smem; float4* Qn = smem + PC;
  const int m0 = blockIdx.y * PC, cntp = min(PC, n - m0), padded = (cntp + 31) & ~31;  // whole pairs of 16-point groups
  for (int t = threadIdx.x; t < padded; t += 256) {
    const int m = m0 + t;
    if (t < cntp) { P4[t] = make_float4(planes[m], planes[ld + m], planes[2 * ld + m], 1.0f); Qn[t] = make_float4(-planes[3 * ld + m], -planes[4 * ld + m], -planes[5 * ld + m], 0.f); }
    else { P4[t] = make_float4(0, 0, 0, 1.0f); Qn[t] = make_float4(-1e30f, -1e30f, -1e30f, 0.f); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t hyp0 = (blockIdx.x * 4 + wave) * (4 * QUADS);
  const int a_hq = (lane & 15) >> 2, a_c = lane & 3, a_k = lane >> 4;
  float a[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const uint32_t h = hyp0 + 4 * q + a_hq;
    a[q] = a_c < 3 ? Rt[(size_t)(a_k < 3 ? 3 * a_c + a_k : 9 + a_c) * ldl + h] : 0.0f;
  }
  __syncthreads();
  // vertical counters: a lane counts at most PC / 16 = 32 groups -> 6 planes
  uint64_t pl[QUADS][6];
#pragma unroll
  for (int q = 0; q < QUADS; q++)
#pragma unroll
    for (int b = 0; b < 6; b++) pl[q][b] = 0;
  const float* P4f = reinterpret_cast<const float*>(P4);
  const int col = lane & 15, kb = lane >> 4;
  for (int g = 0; g < padded; g += 32) {
    const float b0 = P4f[(g + col) * 4 + kb], b1 = P4f[(g + 16 + col) * 4 + kb];
    const float4 cq0 = Qn[g + col], cq1 = Qn[g + 16 + col];
    const f32x4 c0 = {cq0.x, cq0.y, cq0.z, cq0.w}, c1 = {cq1.x, cq1.y, cq1.z, cq1.w};
#pragma unroll
    for (int q = 0; q < QUADS; q++) {
      const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b0, c0, 0, 0, 0);
      const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b1, c1, 0, 0, 0);
      const float e0 = fma_(d0[2], d0[2], fma_(d0[1], d0[1], d0[0] * d0[0]));
      const float e1 = fma_(d1[2], d1[2], fma_(d1[1], d1[1], d1[0] * d1[0]));
      const uint64_t x = __ballot(e0 < tau2), y = __ballot(e1 < tau2);
      // x + y + plane0 -> plane0 (sum), carry of weight 2; then ripple the carry
      const uint64_t h = x ^ y;
      uint64_t carry = (x & y) | (h & pl[q][0]);
      pl[q][0] ^= h;
#pragma unroll
      for (int bb = 1; bb < 6; bb++) { const uint64_t t2 = pl[q][bb] & carry; pl[q][bb] ^= carry; carry = t2; }
    }
  }
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    uint32_t c = 0;
#pragma unroll
    for (int bb = 0; bb < 6; bb++) c += (uint32_t)((pl[q][bb] >> lane) & 1ull) << bb;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 16);
    const uint32_t h = hyp0 + 4 * q + (lane >> 4);
    if (col == 0) partial[(size_t)blockIdx.y * ldl + h] = c;
  }
}

static uint64_t checksum(const std::vector<uint32_t>& p, uint32_t ldl, int chunks, uint32_t T) {
  uint64_t s = 0;
  for (uint32_t h = 0; h < T; h++) { uint64_t c = 0; for (int k = 0; k < chunks; k++) c += p[(size_t)k * ldl + h]; s = s * 1000003ull + c; }
  return s;
}

int main() {
  const int n = 5000, ld = 5120; const uint32_t T = 50176, ldl = T; const int chunks = (n + PC - 1) / PC;
  std::vector<float> planes(6 * (size_t)ld, 0.f), Rt(12 * (size_t)ldl);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (int m = 0; m < n; m++) { for (int c = 0; c < 3; c++) planes[c * ld + m] = rnd(); for (int c = 3; c < 6; c++) planes[c * ld + m] = planes[(c - 3) * ld + m] + 0.05f * rnd(); }
  for (uint32_t h = 0; h < T; h++) {  // near-identity transforms: some inliers for every hypothesis
    for (int c = 0; c < 9; c++) Rt[(size_t)c * ldl + h] = (c % 4 == 0 ? 1.f : 0.f) + 0.02f * rnd();
    for (int c = 9; c < 12; c++) Rt[(size_t)c * ldl + h] = 0.03f * rnd();
  }
  float *d_pl, *d_Rt; uint32_t* d_part;
  hipMalloc(&d_pl, planes.size() * 4); hipMalloc(&d_Rt, Rt.size() * 4); hipMalloc(&d_part, (size_t)chunks * ldl * 4);
  hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_Rt, Rt.data(), Rt.size() * 4, hipMemcpyHostToDevice);
  const float tau2 = 0.05f * 0.05f;
  std::vector<uint32_t> part((size_t)chunks * ldl);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto bench = [&](const char* name, auto launch) {
    hipMemset(d_part, 0, part.size() * 4);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(part.data(), d_part, part.size() * 4, hipMemcpyDeviceToHost);
    printf("%-34s %7.1f us   checksum %016llx\n", name, ms * 1000 / 20, (unsigned long long)checksum(part, ldl, chunks, T));
  };
#define L(K, HPB) [&] { hipLaunchKernelGGL(K, dim3(T / (HPB), chunks), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part); }
  bench("valu (lane = hypothesis)", L(k_valu, 256));
  bench("mfma only, 8 quads (rate)", L((k_mfma<8, 0>), 128));
  bench("mfma only, 4 quads (rate)", L((k_mfma<4, 0>), 64));
#define LR(K, HPB, R) [&] { hipLaunchKernelGGL(K, dim3(T / (HPB), chunks), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, R); }
  bench("mfma only, 8 quads, loop x4", LR((k_mfma<8, 0>), 128, 4));
  bench("mfma only, 8 quads, loop x16", LR((k_mfma<8, 0>), 128, 16));
  bench("mfma only, 4 quads, loop x4", LR((k_mfma<4, 0>), 64, 4));
  bench("mfma+epi, 8 quads, loop x4", LR((k_mfma<8, 1>), 128, 4));
  bench("mfma+epi, 8 quads, loop x16", LR((k_mfma<8, 1>), 128, 16));
  bench("mfma+epi, 4 quads, loop x4", LR((k_mfma<4, 1>), 64, 4));
  bench("mfma + valu epilogue, 8 quads", L((k_mfma<8, 1>), 128));
  bench("mfma + valu epilogue, 4 quads", L((k_mfma<4, 1>), 64));
  bench("mfma + valu epilogue, 2 quads", L((k_mfma<2, 1>), 32));
  bench("mfma + salu counters, 8 quads", L((k_mfma_salu<8>), 128));
  bench("mfma + salu counters, 4 quads", L((k_mfma_salu<4>), 64));
  bench("mfma + salu counters, 2 quads", L((k_mfma_salu<2>), 32));
  return 0;
}
