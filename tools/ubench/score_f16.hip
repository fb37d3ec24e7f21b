// Prototype for stage C2: inlier counting with an fp16-split matrix-pipe FILTER and exact fp32 re-evaluation of the
// tests the filter cannot decide.  Must print the same checksum as the plain fp32 kernel.
//   bash tools/ubench/run.sh score_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PC = 512;
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float canon_d2(const float* M, const float* p) {
  const float ex = M[9] + fma_(M[2], p[2], fma_(M[1], p[1], fma_(M[0], p[0], -p[3])));
  const float ey = M[10] + fma_(M[5], p[2], fma_(M[4], p[1], fma_(M[3], p[0], -p[4])));
  const float ez = M[11] + fma_(M[8], p[2], fma_(M[7], p[1], fma_(M[6], p[0], -p[5])));
  return fma_(ez, ez, fma_(ey, ey, ex * ex));
}

__global__ __launch_bounds__(256, 8) void k_valu(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                 uint32_t ldl, float tau2, uint32_t* __restrict__ partial, uint32_t* dbg) {
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
    const float p[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    c0 += canon_d2(M, p) < tau2 ? 1u : 0u;
  }
  partial[(size_t)blockIdx.y * ldl + l] = c0;
}

// ---------------------------------------------------------------------------------------------------------------
// filter kernel.  Workgroup = 4 waves; a wave owns 8 RB hypotheses (RB row blocks of 8); chunk of <= 512 points.
// MFMA 32x32x16 f16: rows = (hypothesis, component) [4 per hypothesis, the 4th idle], columns = 32 points, K = 16:
//   k0..8  : (rh,rh,rl) x (Ph,Pl,Ph) for x, y, z      r = 1024 R (hi + lo halves), P = s p (hi + lo halves)
//   k9..14 : -1024 x (Qh, Ql) of the row's own component
//   C      : 1024 s t (fp32, exact)
// so D = 1024 s (R p + t - q) up to the split and accumulation error; lane (col, hf) holds the three components of
// hypotheses 2 jj + hf (jj = 0..3) for point col: the squared norm and the tests stay lane-local.
// ---------------------------------------------------------------------------------------------------------------
constexpr float RS = 1024.0f;
template <int RB>
__global__ __launch_bounds__(256) void k_f16(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                             uint32_t ldl, float tau2, uint32_t* __restrict__ partial, uint32_t* dbg) {
  constexpr int HB = 4 * RB * 8;  // hypotheses per workgroup
  __shared__ uint4 Bt[PC * 2];
  __shared__ float P32[PC * 6];
  __shared__ float Hc[HB * 12];
  __shared__ float Tabs[HB];
  __shared__ uint32_t wildf[HB];
  __shared__ uint32_t fix[HB];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * PC, cntp = min(PC, n - m0), padded = (cntp + 31) & ~31;
  // --- points: load, chunk maxima
  float v[2][6];
  float mp = 0.f, mq = 0.f;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = tid + 256 * u, m = m0 + t;
    if (t < cntp) {
#pragma unroll
      for (int c = 0; c < 6; c++) v[u][c] = planes[(size_t)c * ld + m];
      mp = fmaxf(mp, fmaxf(fabsf(v[u][0]), fmaxf(fabsf(v[u][1]), fabsf(v[u][2]))));
      mq = fmaxf(mq, fmaxf(fabsf(v[u][3]), fmaxf(fabsf(v[u][4]), fabsf(v[u][5]))));
    } else {
      v[u][0] = v[u][1] = v[u][2] = 0.f; v[u][3] = v[u][4] = v[u][5] = 1e30f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
  if (lane == 0) { red[wave] = mp; red[4 + wave] = mq; }
  if (tid < HB) {  // the workgroup's hypotheses, fp32, for the exact re-evaluation
    const uint32_t h = blockIdx.x * HB + tid;
    float ta = 0.f; bool w = false;
#pragma unroll
    for (int c = 0; c < 12; c++) {
      const float x = Rt[(size_t)c * ldl + h];
      Hc[tid * 12 + c] = x;
      if (c < 9) w = w || !(fabsf(x) <= 1.5f); else { ta = fmaxf(ta, fabsf(x)); w = w || !(fabsf(x) < 1e30f); }
    }
    Tabs[tid] = ta; wildf[tid] = w ? 1u : 0u; fix[tid] = 0u;
  }
  __syncthreads();
  const float Pmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), Qmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float mx = fmaxf(Pmax, Qmax);
  int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;  // mx in [2^e, 2^(e+1))
  int k = 8 - e; k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float s = __uint_as_float((uint32_t)(k + 127) << 23);
  // --- B tile (fp16 hi / lo splits of the scaled coordinates) and the fp32 copy
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = tid + 256 * u;
    if (t < padded) {
      _Float16 hi[6], lo[6];
      const bool real = t < cntp;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const float X = real ? v[u][c] * s : (c < 3 ? 0.f : 32768.f);
        hi[c] = (_Float16)X; lo[c] = (_Float16)(X - (float)hi[c]);
      }
      half8 f0 = {hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2]};
      half8 f1 = {hi[2], hi[3], lo[3], hi[4], lo[4], hi[5], lo[5], (_Float16)0.f};
      Bt[t * 2] = *reinterpret_cast<uint4*>(&f0); Bt[t * 2 + 1] = *reinterpret_cast<uint4*>(&f1);
#pragma unroll
      for (int c = 0; c < 6; c++) P32[t * 6 + c] = v[u][c];
    }
  }
  // --- per wave: error bound, thresholds
  const int hw0 = wave * 8 * RB;  // first local hypothesis of this wave
  float tmax = 0.f;
  if (lane < 8 * RB && !wildf[hw0 + lane]) tmax = Tabs[hw0 + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  const float st = s * sqrtf(tau2);
  const float Sb = 2.6f * (Pmax * s) + Qmax * s + tmax * s;
  const float eta = Sb * (1.0f / 65536.0f);
  const bool fast = (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f);  // false on NaN
  const float lo_e = RS * (st - eta), hi_e = RS * (st + eta);
  const float LO = lo_e * lo_e * (1.0f - 1e-6f), HI = hi_e * hi_e * (1.0f + 1e-6f);
  const uint32_t W2b = __float_as_uint(HI - LO);
  const int col = lane & 31, hf = lane >> 5;
  // --- A fragments and C tiles
  half8 A[RB];
  f32x16 C[RB];
  uint32_t wl[RB];  // bit jj: hypothesis (rb, jj) of this lane is wild
#pragma unroll
  for (int rb = 0; rb < RB; rb++) {
    const int hy = (lane & 31) >> 2, c = lane & 3;
    const float* M = &Hc[(hw0 + 8 * rb + hy) * 12];
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const float x = c < 3 ? M[3 * c + kk] * RS : 0.f;
      rh[kk] = (_Float16)x; rl[kk] = (_Float16)(x - (float)rh[kk]);
    }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A[rb] = hf ? a1 : a0;
    wl[rb] = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const int hl = hw0 + 8 * rb + 2 * jj + hf;
      const float* T = &Hc[hl * 12 + 9];
      C[rb][4 * jj] = T[0] * s * RS; C[rb][4 * jj + 1] = T[1] * s * RS; C[rb][4 * jj + 2] = T[2] * s * RS; C[rb][4 * jj + 3] = 0.f;
      wl[rb] |= (wildf[hl] || !fast) ? (1u << jj) : 0u;
    }
  }
  __syncthreads();
  uint32_t sr[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) sr[rb][jj] = 0;
  const half8* Bh = reinterpret_cast<const half8*>(Bt);
  uint32_t events = 0;
  for (int g = 0; g < padded; g += 32) {
    const half8 b = Bh[(g + col) * 2 + hf];
    float x[RB][4];
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
      const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[rb], b, C[rb], 0, 0, 0);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        x[rb][jj] = fma_(D[4 * jj + 2], D[4 * jj + 2], fma_(D[4 * jj + 1], D[4 * jj + 1], fma_(D[4 * jj], D[4 * jj], -LO)));
        sr[rb][jj] = __builtin_amdgcn_alignbit(sr[rb][jj], __float_as_uint(x[rb][jj]), 31);
        mn = min(mn, __float_as_uint(x[rb][jj]));
      }
    }
    if (__ballot(mn < W2b)) {  // some test of this batch lies in the undecided shell: settle it exactly
#pragma unroll
      for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
          if (__float_as_uint(x[rb][jj]) < W2b && !((wl[rb] >> jj) & 1u)) {
            const int hl = hw0 + 8 * rb + 2 * jj + hf;
            events++;
            if (canon_d2(&Hc[hl * 12], &P32[(g + col) * 6]) < tau2) atomicAdd(&fix[hl], 1u);
          }
    }
  }
  // --- wild hypotheses (or a chunk the filter does not cover): this lane's 16 tests, exactly
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
      if ((wl[rb] >> jj) & 1u) {
        const int hl = hw0 + 8 * rb + 2 * jj + hf;
        uint32_t bits = 0;
        for (int g = 0; g < padded; g += 32) bits = (bits << 1) | (canon_d2(&Hc[hl * 12], &P32[(g + col) * 6]) < tau2 ? 1u : 0u);
        sr[rb][jj] = bits;
      }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      uint32_t c = (uint32_t)__popc(sr[rb][jj]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
      const int hl = hw0 + 8 * rb + 2 * jj + hf;
      if (col == 0) partial[(size_t)blockIdx.y * ldl + blockIdx.x * HB + hl] = c + fix[hl];
    }
  if (dbg && events) atomicAdd(dbg, events);
}


// ---------------------------------------------------------------------------------------------------------------
// v2: the fp16 tile of the points is made ONCE (prep kernel, 32 B per point + one scale and the maxima per chunk of
// 512); the scoring kernel reads its B operand straight from global memory (1 KiB contiguous per wave and MFMA step,
// L1/L2 resident), keeps its A fragments for a whole range of chunks, and queues undecided tests in LDS for a
// lane-parallel exact pass.
// ---------------------------------------------------------------------------------------------------------------
struct ChunkInfo { float s, pmax, qmax, pad; };

__global__ __launch_bounds__(256) void k_prep(const float* __restrict__ planes, int n, int ld, uint4* __restrict__ tile,
                                              ChunkInfo* __restrict__ info) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * PC, cntp = min(PC, n - m0);
  float v[2][6];
  float mp = 0.f, mq = 0.f;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = tid + 256 * u, m = m0 + t;
    if (t < cntp) {
#pragma unroll
      for (int c = 0; c < 6; c++) v[u][c] = planes[(size_t)c * ld + m];
      mp = fmaxf(mp, fmaxf(fabsf(v[u][0]), fmaxf(fabsf(v[u][1]), fabsf(v[u][2]))));
      mq = fmaxf(mq, fmaxf(fabsf(v[u][3]), fmaxf(fabsf(v[u][4]), fabsf(v[u][5]))));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
  if (lane == 0) { red[wave] = mp; red[4 + wave] = mq; }
  __syncthreads();
  const float Pmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), Qmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float mx = fmaxf(Pmax, Qmax);
  int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 127;
  int k = 8 - e; k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float s = __uint_as_float((uint32_t)(k + 127) << 23);
  if (tid == 0) info[blockIdx.x] = ChunkInfo{s, Pmax, Qmax, 0.f};
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = tid + 256 * u;
    _Float16 hi[6], lo[6];
    const bool real = t < cntp;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const float X = real ? v[u][c] * s : (c < 3 ? 0.f : 32768.f);
      hi[c] = (_Float16)X; lo[c] = (_Float16)(X - (float)hi[c]);
    }
    half8 f0 = {hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2]};
    half8 f1 = {hi[2], hi[3], lo[3], hi[4], lo[4], hi[5], lo[5], (_Float16)0.f};
    tile[((size_t)m0 + t) * 2] = *reinterpret_cast<uint4*>(&f0); tile[((size_t)m0 + t) * 2 + 1] = *reinterpret_cast<uint4*>(&f1);
  }
}

constexpr int QCAP = 256;
template <int RB, int ABL = 0>
__global__ __launch_bounds__(256) void k_f16v2(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                               uint32_t ldl, float tau2, uint32_t* __restrict__ cnt_out, uint32_t* dbg,
                                               const uint4* __restrict__ tile, const ChunkInfo* __restrict__ info, int chunks,
                                               int splits) {
  constexpr int HB = 4 * RB * 8;
  __shared__ float Hc[HB * 12];
  __shared__ float Tabs[HB];
  __shared__ uint32_t wildf[HB];
  __shared__ uint32_t fix[HB];
  __shared__ uint32_t queue[4][QCAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < HB) {
    const uint32_t h = blockIdx.x * HB + tid;
    float ta = 0.f; bool w = false;
#pragma unroll
    for (int c = 0; c < 12; c++) {
      const float x = Rt[(size_t)c * ldl + h];
      Hc[tid * 12 + c] = x;
      if (c < 9) w = w || !(fabsf(x) <= 1.5f); else { ta = fmaxf(ta, fabsf(x)); w = w || !(fabsf(x) < 1e30f); }
    }
    Tabs[tid] = ta; wildf[tid] = w ? 1u : 0u; fix[tid] = 0u;
  }
  __syncthreads();
  const int hw0 = wave * 8 * RB;
  float tmax = 0.f;
  if (lane < 8 * RB && !wildf[hw0 + lane]) tmax = Tabs[hw0 + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  const int col = lane & 31, hf = lane >> 5;
  half8 A[RB];
  uint32_t wl[RB];
#pragma unroll
  for (int rb = 0; rb < RB; rb++) {
    const int hy = (lane & 31) >> 2, c = lane & 3;
    const float* M = &Hc[(hw0 + 8 * rb + hy) * 12];
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const float x = c < 3 ? M[3 * c + kk] * RS : 0.f;
      rh[kk] = (_Float16)x; rl[kk] = (_Float16)(x - (float)rh[kk]);
    }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A[rb] = hf ? a1 : a0;
    wl[rb] = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) wl[rb] |= wildf[hw0 + 8 * rb + 2 * jj + hf] ? (1u << jj) : 0u;
  }
  uint32_t total[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) total[rb][jj] = 0;
  uint32_t qn = 0, events = 0;  // wave-uniform queue fill
  uint32_t* q = queue[wave];
  const float* aos = planes;  // exact pass reads the SoA planes
  auto drain = [&]() {
    for (uint32_t i = lane; i < qn; i += 64) {
      const uint32_t ent = q[i];
      const int hl = ent >> 16, m = ent & 0xFFFF;  // m: point index inside the call's chunk range, see below
      float p[6];
#pragma unroll
      for (int c = 0; c < 6; c++) p[c] = aos[(size_t)c * ld + m];
      if (canon_d2(&Hc[hl * 12], p) < tau2) atomicAdd(&fix[hl], 1u);
    }
    qn = 0;
  };
  const half8* Th = reinterpret_cast<const half8*>(tile);
  for (int ch = blockIdx.y; ch < chunks; ch += splits) {
    const ChunkInfo ci = info[ch];
    const int m0 = ch * PC, cntp = min(PC, n - m0), padded = (cntp + 31) & ~31;
    const float s = ci.s, st = s * sqrtf(tau2);
    const float Sb = 2.6f * (ci.pmax * s) + ci.qmax * s + tmax * s;
    const float eta = Sb * (1.0f / 65536.0f);
    const bool fast = (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f);
    const float lo_e = RS * (st - eta), hi_e = RS * (st + eta);
    const float LO = lo_e * lo_e * (1.0f - 1e-6f), HI = hi_e * hi_e * (1.0f + 1e-6f);
    const uint32_t W2b = __float_as_uint(HI - LO);
    f32x16 C[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const float* T = &Hc[(hw0 + 8 * rb + 2 * jj + hf) * 12 + 9];
        C[rb][4 * jj] = T[0] * s * RS; C[rb][4 * jj + 1] = T[1] * s * RS; C[rb][4 * jj + 2] = T[2] * s * RS; C[rb][4 * jj + 3] = 0.f;
      }
    uint32_t sr[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
      for (int jj = 0; jj < 4; jj++) sr[rb][jj] = 0;
    const half8* Bp = Th + ((size_t)m0 + col) * 2 + hf;
    half8 bn = Bp[0];
    for (int g = 0; g < padded; g += 32) {
      const half8 b = bn;
      if (!(ABL & 1) && g + 32 < padded) bn = Bp[(size_t)(g + 32) * 2];
      float x[RB][4];
      uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
      for (int rb = 0; rb < RB; rb++) {
        f32x16 D;
        if (ABL & 2) { D = C[rb]; D[0] += (float)b[0]; D[5] += (float)b[1]; D[10] += (float)b[2]; D[12] += (float)b[3]; }
        else D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[rb], b, C[rb], 0, 0, 0);
        if (ABL & 8) {  // no epilogue: fold the tile into one value
          float acc = 0.f;
#pragma unroll
          for (int r = 0; r < 16; r += 5) acc += D[r];
          sr[rb][0] += __float_as_uint(acc);
        } else {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          x[rb][jj] = fma_(D[4 * jj + 2], D[4 * jj + 2], fma_(D[4 * jj + 1], D[4 * jj + 1], fma_(D[4 * jj], D[4 * jj], -LO)));
          sr[rb][jj] = __builtin_amdgcn_alignbit(sr[rb][jj], __float_as_uint(x[rb][jj]), 31);
          mn = min(mn, __float_as_uint(x[rb][jj]));
        }
        }
      }
      if (!(ABL & 12) && fast && __ballot(mn < W2b)) {
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const bool hit = __float_as_uint(x[rb][jj]) < W2b && !((wl[rb] >> jj) & 1u);
            const uint64_t hm = __ballot(hit);
            if (hm) {
              if (qn + 64 > QCAP) drain();
              if (hit) q[qn + __popcll(hm & ((1ull << lane) - 1ull))] = ((uint32_t)(hw0 + 8 * rb + 2 * jj + hf) << 16) | (uint32_t)(m0 + g + col);
              const uint32_t k2 = (uint32_t)__popcll(hm);
              qn += k2; events += k2;
            }
          }
      }
    }
    // wild hypotheses, or a chunk outside the filter's range: this lane's tests of the chunk, exactly
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        if (!fast || ((wl[rb] >> jj) & 1u)) {
          const int hl = hw0 + 8 * rb + 2 * jj + hf;
          uint32_t bits = 0;
          for (int g = 0; g < padded; g += 32) {
            const int m = m0 + g + col;
            float p[6];
#pragma unroll
            for (int c = 0; c < 6; c++) p[c] = m < n ? planes[(size_t)c * ld + m] : (c < 3 ? 0.f : 1e30f);
            bits = (bits << 1) | (canon_d2(&Hc[hl * 12], p) < tau2 ? 1u : 0u);
          }
          sr[rb][jj] = bits;
        }
        total[rb][jj] += (uint32_t)__popc(sr[rb][jj]);
      }
  }
  drain();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
  for (int rb = 0; rb < RB; rb++)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      uint32_t c = total[rb][jj];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
      const int hl = hw0 + 8 * rb + 2 * jj + hf;
      if (col == 0) cnt_out[(size_t)blockIdx.y * ldl + blockIdx.x * HB + hl] = c + fix[hl];
    }
  if (dbg && lane == 0 && events) atomicAdd(dbg, events);
}


// (v3 .. v6 — the exact pass inside the filter kernel in four shapes: per-batch enqueue + drain through a non-inlined
//  slow path, a double-buffered chunk loop, a 1280-entry workgroup queue, one row block per wave with the constants a chunk
//  ahead — were removed from this file in round 3; their measurements are the "v3 .. v6" rows of
//  profiles/r02_ubench_score_filter_prototype.txt and DESIGN.md §5: every one of them paid 27-50 us for cold code inside
//  the hot loop.  What remains: the fp32 kernel, the first filter (v1), the tile + straight-from-global form with its
//  ablations (v2), and the two LDS-DMA forms that led to the product kernel (v7, v8).)

constexpr int Q7 = 640;
template <int WAVES, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES, 6) void k_f16v7(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                          uint32_t ldl, float tau2, uint32_t* __restrict__ cnt_out, uint32_t* dbg,
                                                          const uint4* __restrict__ tile, const ChunkInfo* __restrict__ info, int chunks,
                                                          int splits) {
  __shared__ uint4 Bt[PC * 2];
  __shared__ float4 Ttab[WAVES][8];  // per wave: translation (x 1024) and the wild flag of its 8 hypotheses
  __shared__ uint32_t fix[WAVES][8];
  __shared__ uint32_t queue[WAVES][Q7];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hf = lane >> 5;
  const int per = (chunks + splits - 1) / splits, c0 = blockIdx.y * per, c1 = min(chunks, c0 + per);
  const uint32_t h0 = (blockIdx.x * WAVES + wave) * 8;  // this wave's hypotheses
  // --- A fragment: row r = (hypothesis r >> 2, component r & 3)
  half8 A;
  {
    const int r = lane & 31, hy = r >> 2, c = r & 3;
    float x[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) x[kk] = c < 3 ? Rt[(size_t)(3 * c + kk) * ldl + h0 + hy] * RS : 0.f;
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { rh[kk] = (_Float16)x[kk]; rl[kk] = (_Float16)(x[kk] - (float)rh[kk]); }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A = hf ? a1 : a0;
  }
  float tmax = 0.f;
  if (lane < 8) {
    const uint32_t h = h0 + lane;
    bool w = false; float ta = 0.f, t[3];
#pragma unroll
    for (int c = 0; c < 9; c++) w = w || !(fabsf(Rt[(size_t)c * ldl + h]) <= 1.5f);
#pragma unroll
    for (int c = 0; c < 3; c++) { t[c] = Rt[(size_t)(9 + c) * ldl + h]; ta = fmaxf(ta, fabsf(t[c])); w = w || !(fabsf(t[c]) < 1e30f); }
    Ttab[wave][lane] = make_float4(t[0] * RS, t[1] * RS, t[2] * RS, w ? 1.f : 0.f);
    fix[wave][lane] = 0u;
    tmax = w ? 0.f : ta;
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  tmax = __shfl(tmax, 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t wl = 0;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) wl |= Ttab[wave][2 * jj + hf].w != 0.f ? (1u << jj) : 0u;
  uint32_t total[4] = {0, 0, 0, 0};
  uint32_t qn = 0, events = 0;
  uint32_t* q = queue[wave];
  auto drain = [&]() {
    for (uint32_t i = lane; i < qn; i += 64) {
      const uint32_t ent = q[i];
      const int m = (int)(ent >> 5), ehf = (ent >> 4) & 1;
      float p[6];
#pragma unroll
      for (int c = 0; c < 6; c++) p[c] = planes[(size_t)c * ld + m];
      for (uint32_t bits = ent & 0xFu; bits; bits &= bits - 1) {
        const int hl = 2 * (__ffs(bits) - 1) + ehf;
        float M[12];
#pragma unroll
        for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + h0 + hl];
        if (canon_d2(M, p) < tau2) atomicAdd(&fix[wave][hl], 1u);
      }
    }
    events += qn; qn = 0;
  };
  const half8* Bc = reinterpret_cast<const half8*>(Bt) + col * 2 + hf;  // step g: Bc[64 g]
  for (int ch = c0; ch < c1; ch++) {
    if (!(ABL & 1) || ch == c0) {
    __syncthreads();  // every wave is done with the previous tile
#pragma unroll
    for (int i = 0; i < 16 / WAVES; i++)  // 1024 x 16 bytes by 64 WAVES lanes
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + (size_t)ch * (PC * 2) + 64 * WAVES * i + tid),
                                       (__attribute__((address_space(3))) void*)(&Bt[64 * WAVES * i + wave * 64]), 16, 0, 0);
    }
    const ChunkInfo ci = info[ch];
    const int m0 = ch * PC;
    const float s = ci.s, st = s * sqrtf(tau2);
    const float Sb = 2.6f * (ci.pmax * s) + ci.qmax * s + tmax * s;
    const float eta = Sb * (1.0f / 65536.0f);
    const bool fast = (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f);
    const float lo_e = RS * (st - eta), hi_e = RS * (st + eta);
    const float LO = lo_e * lo_e * (1.0f - 1e-6f), HI = hi_e * hi_e * (1.0f + 1e-6f);
    const uint32_t W2b = fast ? __float_as_uint(HI - LO) : 0u;
    f32x16 C;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const float4 T = Ttab[wave][2 * jj + hf];
      C[4 * jj] = T.x * s; C[4 * jj + 1] = T.y * s; C[4 * jj + 2] = T.z * s; C[4 * jj + 3] = 0.f;
    }
    uint32_t sr[4] = {0, 0, 0, 0};
    if (!(ABL & 1) || ch == c0) __syncthreads();  // the tile has landed (the barrier waits for the DMA)
    half8 b = Bc[0];
#pragma unroll 2
    for (int g = 0; g < 16; g++) {
      const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
      if (g + 1 < 16) b = Bc[64 * (g + 1)];
      float x[4];
      uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const float v = fma_(D[4 * jj + 2], D[4 * jj + 2], fma_(D[4 * jj + 1], D[4 * jj + 1], fma_(D[4 * jj], D[4 * jj], -LO)));
        x[jj] = v;
        sr[jj] = __builtin_amdgcn_alignbit(sr[jj], __float_as_uint(v), 31);
        mn = min(mn, __float_as_uint(v));
      }
      const uint64_t hm = (ABL & 2) ? 0ull : __ballot(mn < W2b);
      if (__builtin_expect(hm != 0, 0)) {
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) bits |= (__float_as_uint(x[t]) < W2b) ? (1u << t) : 0u;
        bits &= ~wl;
        if (mn < W2b) q[qn + __popcll(hm & ((1ull << lane) - 1ull))] = ((uint32_t)(m0 + 32 * g + col) << 5) | ((uint32_t)hf << 4) | bits;
        qn += (uint32_t)__popcll(hm);
        if (qn > Q7 - 64) drain();
      }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!(ABL & 4) && (!fast || ((wl >> t) & 1u))) {
        const int hl = 2 * t + hf;
        float M[12];
#pragma unroll
        for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + h0 + hl];
        uint32_t bits = 0;
        for (int g = 0; g < 16; g++) {
          const int m = m0 + 32 * g + col;
          float p[6];
#pragma unroll
          for (int c = 0; c < 6; c++) p[c] = m < n ? planes[(size_t)c * ld + m] : (c < 3 ? 0.f : 1e30f);
          bits = (bits << 1) | (canon_d2(M, p) < tau2 ? 1u : 0u);
        }
        sr[t] = bits;
      }
      total[t] += (uint32_t)__popc(sr[t]);
    }
  }
  drain();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t c = total[t];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
    const int hl = 2 * t + hf;
    if (col == 0) cnt_out[(size_t)blockIdx.y * ldl + h0 + hl] = c + fix[wave][hl];
  }
  if (dbg && lane == 0 && events) atomicAdd(dbg, events);
}


// ---------------------------------------------------------------------------------------------------------------
// v8: v7 with (a) the B tile double buffered, the DMA of chunk c+1 in flight during chunk c, ONE barrier per chunk;
// (b) everything cold (exact pass over the queue, exact recount of a wave's tests) in a function that is called at chunk
// boundaries only; (c) a small queue that saturates: a wave whose queue overflows recounts the chunk exactly.
// ---------------------------------------------------------------------------------------------------------------
constexpr int Q8 = 256;
struct Cold {
  const float* planes; const float* Rt; int ld; uint32_t ldl; int n; float tau2;
};
// exact pass over the queued tests (entries: point << 5 | lane half << 4 | tests), counts added to fix[0..7]
__device__ __noinline__ void cold_drain(const Cold& k, uint32_t h0, const uint32_t* q, uint32_t qn, uint32_t* fix) {
  const int lane = threadIdx.x & 63;
  for (uint32_t i = lane; i < qn; i += 64) {
    const uint32_t ent = q[i];
    const int m = (int)(ent >> 5), ehf = (ent >> 4) & 1;
    float p[6];
#pragma unroll
    for (int c = 0; c < 6; c++) p[c] = k.planes[(size_t)c * k.ld + m];
    for (uint32_t bits = ent & 0xFu; bits; bits &= bits - 1) {
      const int hl = 2 * (__ffs(bits) - 1) + ehf;
      float M[12];
#pragma unroll
      for (int c = 0; c < 12; c++) M[c] = k.Rt[(size_t)c * k.ldl + h0 + hl];
      if (canon_d2(M, p) < k.tau2) atomicAdd(&fix[hl], 1u);
    }
  }
}
// this lane's 16 tests of one chunk for hypothesis h, exactly: the bit pattern the filter would have left in sr
__device__ __noinline__ uint32_t cold_recount(const Cold& k, uint32_t h, int m0) {
  const int col = threadIdx.x & 31;
  float M[12];
#pragma unroll
  for (int c = 0; c < 12; c++) M[c] = k.Rt[(size_t)c * k.ldl + h];
  uint32_t bits = 0;
  for (int g = 0; g < 16; g++) {
    const int m = m0 + 32 * g + col;
    float p[6];
#pragma unroll
    for (int c = 0; c < 6; c++) p[c] = m < k.n ? k.planes[(size_t)c * k.ld + m] : (c < 3 ? 0.f : 1e30f);
    bits = (bits << 1) | (canon_d2(M, p) < k.tau2 ? 1u : 0u);
  }
  return bits;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 6) void k_f16v8(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                                          uint32_t ldl, float tau2, uint32_t* __restrict__ cnt_out, uint32_t* dbg,
                                                          const uint4* __restrict__ tile, const ChunkInfo* __restrict__ info, int chunks,
                                                          int splits) {
  __shared__ uint4 Bt[2][PC * 2];
  __shared__ float4 Ttab[WAVES][8];
  __shared__ uint32_t fix[WAVES][8];
  __shared__ uint32_t queue[WAVES][Q8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hf = lane >> 5;
  const int per = (chunks + splits - 1) / splits, c0 = blockIdx.y * per, c1 = min(chunks, c0 + per);
  const uint32_t h0 = (blockIdx.x * WAVES + wave) * 8;
  auto stage = [&](int ch, int buf) {
#pragma unroll
    for (int i = 0; i < 16 / WAVES; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile + (size_t)ch * (PC * 2) + 64 * WAVES * i + tid),
                                       (__attribute__((address_space(3))) void*)(&Bt[buf][64 * WAVES * i + wave * 64]), 16, 0, 0);
  };
  if (c0 < c1) stage(c0, 0);
  half8 A;
  {
    const int r = lane & 31, hy = r >> 2, c = r & 3;
    float x[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) x[kk] = c < 3 ? Rt[(size_t)(3 * c + kk) * ldl + h0 + hy] * RS : 0.f;
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { rh[kk] = (_Float16)x[kk]; rl[kk] = (_Float16)(x[kk] - (float)rh[kk]); }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A = hf ? a1 : a0;
  }
  float tmax = 0.f;
  if (lane < 8) {
    const uint32_t h = h0 + lane;
    bool w = false; float ta = 0.f, t[3];
#pragma unroll
    for (int c = 0; c < 9; c++) w = w || !(fabsf(Rt[(size_t)c * ldl + h]) <= 1.5f);
#pragma unroll
    for (int c = 0; c < 3; c++) { t[c] = Rt[(size_t)(9 + c) * ldl + h]; ta = fmaxf(ta, fabsf(t[c])); w = w || !(fabsf(t[c]) < 1e30f); }
    Ttab[wave][lane] = make_float4(t[0] * RS, t[1] * RS, t[2] * RS, w ? 1.f : 0.f);
    fix[wave][lane] = 0u;
    tmax = w ? 0.f : ta;
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  tmax = __shfl(tmax, 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t wl = 0;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) wl |= Ttab[wave][2 * jj + hf].w != 0.f ? (1u << jj) : 0u;
  uint32_t total[4] = {0, 0, 0, 0};
  uint32_t qn = 0, events = 0;
  uint32_t* q = queue[wave];
  const Cold cold{planes, Rt, ld, ldl, n, tau2};
  for (int ch = c0; ch < c1; ch++) {
    const int buf = (ch - c0) & 1;
    __syncthreads();  // chunk ch's tile has landed; every wave has left the other buffer
    if (ch + 1 < c1) stage(ch + 1, buf ^ 1);
    const ChunkInfo ci = info[ch];
    const int m0 = ch * PC;
    const float s = ci.s, st = s * sqrtf(tau2);
    const float Sb = 2.6f * (ci.pmax * s) + ci.qmax * s + tmax * s;
    const float eta = Sb * (1.0f / 65536.0f);
    const bool fast = (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f);
    const float lo_e = RS * (st - eta), hi_e = RS * (st + eta);
    const float LO = lo_e * lo_e * (1.0f - 1e-6f), HI = hi_e * hi_e * (1.0f + 1e-6f);
    const uint32_t W2b = fast ? __float_as_uint(HI - LO) : 0u;
    f32x16 C;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const float4 T = Ttab[wave][2 * jj + hf];
      C[4 * jj] = T.x * s; C[4 * jj + 1] = T.y * s; C[4 * jj + 2] = T.z * s; C[4 * jj + 3] = 0.f;
    }
    uint32_t sr[4] = {0, 0, 0, 0};
    const uint32_t qn0 = qn;
    bool over = false;
    const half8* Bc = reinterpret_cast<const half8*>(Bt[buf]) + col * 2 + hf;
    half8 b = Bc[0];
#pragma unroll 2
    for (int g = 0; g < 16; g++) {
      const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
      if (g + 1 < 16) b = Bc[64 * (g + 1)];
      float x[4];
      uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const float v = fma_(D[4 * jj + 2], D[4 * jj + 2], fma_(D[4 * jj + 1], D[4 * jj + 1], fma_(D[4 * jj], D[4 * jj], -LO)));
        x[jj] = v;
        sr[jj] = __builtin_amdgcn_alignbit(sr[jj], __float_as_uint(v), 31);
        mn = min(mn, __float_as_uint(v));
      }
      const uint64_t hm = __ballot(mn < W2b);
      if (__builtin_expect(hm != 0, 0)) {
        const uint32_t k2 = (uint32_t)__popcll(hm);
        if (qn + k2 <= Q8) {
          uint32_t bits = 0;
#pragma unroll
          for (int t = 0; t < 4; t++) bits |= (__float_as_uint(x[t]) < W2b) ? (1u << t) : 0u;
          bits &= ~wl;
          if (mn < W2b) q[qn + __popcll(hm & ((1ull << lane) - 1ull))] = ((uint32_t)(m0 + 32 * g + col) << 5) | ((uint32_t)hf << 4) | bits;
          qn += k2;
        } else over = true;
      }
    }
    if (over) qn = qn0;  // the queue overflowed during this chunk: its entries are dropped, the chunk is recounted exactly
    if (__builtin_expect(over || !fast || wl != 0, 0)) {
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (over || !fast || ((wl >> t) & 1u)) sr[t] = cold_recount(cold, h0 + 2 * t + hf, m0);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) total[t] += (uint32_t)__popc(sr[t]);
    if (__builtin_expect(qn > Q8 / 2, 0)) { cold_drain(cold, h0, q, qn, fix[wave]); events += qn; qn = 0; }
  }
  if (qn) { cold_drain(cold, h0, q, qn, fix[wave]); events += qn; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t c = total[t];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
    const int hl = 2 * t + hf;
    if (col == 0) cnt_out[(size_t)blockIdx.y * ldl + h0 + hl] = c + fix[wave][hl];
  }
  if (dbg && lane == 0 && events) atomicAdd(dbg, events);
}

static uint64_t checksum(const std::vector<uint32_t>& p, uint32_t ldl, int chunks, uint32_t T) {
  uint64_t s = 0;
  for (uint32_t h = 0; h < T; h++) { uint64_t c = 0; for (int k = 0; k < chunks; k++) c += p[(size_t)k * ldl + h]; s = s * 1000003ull + c; }
  return s;
}

int main(int argc, char** argv) {
  const int n = 5000, ld = 5120; const uint32_t T = 50176, ldl = T; const int chunks = (n + PC - 1) / PC;
  const float L = argc > 1 ? atof(argv[1]) : 3.0f, tau = argc > 2 ? atof(argv[2]) : 0.1f;
  std::vector<float> planes(6 * (size_t)ld, 0.f), Rt(12 * (size_t)ldl);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  // ground truth rotation about z + translation; 15 % inliers with noise tau / 3, the rest uniform
  const float ang = 0.7f, cg = cosf(ang), sg = sinf(ang), tg[3] = {0.3f * L, -0.2f * L, 0.1f * L};
  for (int m = 0; m < n; m++) {
    float p[3] = {rnd() * L / 2, rnd() * L / 2, rnd() * L / 2};
    float q[3] = {cg * p[0] - sg * p[1] + tg[0], sg * p[0] + cg * p[1] + tg[1], p[2] + tg[2]};
    if (m % 7 != 0) for (int c = 0; c < 3; c++) q[c] = rnd() * L; else for (int c = 0; c < 3; c++) q[c] += tau / 3 * rnd();
    for (int c = 0; c < 3; c++) { planes[c * ld + m] = p[c]; planes[(3 + c) * ld + m] = q[c]; }
  }
  for (uint32_t h = 0; h < T; h++) {  // hypotheses: the truth perturbed by a small rotation about z and a small shift
    const float a = ang + 0.02f * rnd(), ch = cosf(a), sh = sinf(a);
    const float R[9] = {ch, -sh, 0, sh, ch, 0, 0, 0, 1};
    for (int c = 0; c < 9; c++) Rt[(size_t)c * ldl + h] = R[c];
    for (int c = 0; c < 3; c++) Rt[(size_t)(9 + c) * ldl + h] = tg[c] + 0.5f * tau * rnd();
  }
  float *d_pl, *d_Rt; uint32_t *d_part, *d_dbg;
  hipMalloc(&d_pl, planes.size() * 4); hipMalloc(&d_Rt, Rt.size() * 4); hipMalloc(&d_part, (size_t)chunks * ldl * 4); hipMalloc(&d_dbg, 4);
  hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_Rt, Rt.data(), Rt.size() * 4, hipMemcpyHostToDevice);
  const float tau2 = tau * tau;
  std::vector<uint32_t> part((size_t)chunks * ldl);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* only = argc > 3 ? argv[3] : nullptr;
  auto bench = [&](const char* name, auto launch) {
    if (only && !strstr(name, only)) return;
    hipMemset(d_part, 0, part.size() * 4);
    for (int i = 0; i < 3; i++) launch();
    hipMemset(d_dbg, 0, 4);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(part.data(), d_part, part.size() * 4, hipMemcpyDeviceToHost);
    uint32_t ev; hipMemcpy(&ev, d_dbg, 4, hipMemcpyDeviceToHost);
    uint64_t tot = 0; for (uint32_t h = 0; h < T; h++) for (int k = 0; k < chunks; k++) tot += part[(size_t)k * ldl + h];
    printf("%-34s %7.1f us   checksum %016llx  mean inliers %.1f  undecided tests %.4f %%\n", name, ms * 1000 / 20,
           (unsigned long long)checksum(part, ldl, chunks, T), (double)tot / T, 100.0 * ev / 20 / ((double)T * n));
  };
#define L(K, HPB) [&] { hipLaunchKernelGGL(K, dim3(T / (HPB), chunks), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg); }
  printf("L = %g tau = %g\n", L, tau);
  bench("valu fp32 (lane = hypothesis)", L(k_valu, 256));
  bench("f16 filter, 1 row block / wave", L((k_f16<1>), 32));
  bench("f16 filter, 2 row blocks / wave", L((k_f16<2>), 64));
  bench("f16 filter, 4 row blocks / wave", L((k_f16<4>), 128));
  uint4* d_tile; ChunkInfo* d_info;
  hipMalloc(&d_tile, (size_t)(chunks + 2) * PC * 32); hipMemset(d_tile, 0, (size_t)(chunks + 2) * PC * 32); hipMalloc(&d_info, chunks * sizeof(ChunkInfo));
  hipLaunchKernelGGL(k_prep, dim3(chunks), dim3(256), 0, 0, d_pl, n, ld, d_tile, d_info);
  bench("prep (tile of the points)", [&] { hipLaunchKernelGGL(k_prep, dim3(chunks), dim3(256), 0, 0, d_pl, n, ld, d_tile, d_info); });
#define L2(RB, SPL) [&] { hipLaunchKernelGGL((k_f16v2<RB>), dim3(T / (32 * RB), SPL), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg, d_tile, d_info, chunks, SPL); }
  // partial rows beyond `splits` keep old values: clear between variants (bench() memsets)
#define L2A(RB, SPL, AB) [&] { hipLaunchKernelGGL((k_f16v2<RB, AB>), dim3(T / (32 * RB), SPL), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg, d_tile, d_info, chunks, SPL); }
  bench("v2 RB=2 s=1 no B loads", L2A(2, 1, 1));
  bench("v2 RB=2 s=1 no MFMA", L2A(2, 1, 2));
  bench("v2 RB=2 s=1 no event check", L2A(2, 1, 4));
  bench("v2 RB=2 s=1 no epilogue", L2A(2, 1, 8));
  bench("v2 RB=2 s=1 no loads no epilogue", L2A(2, 1, 9));
  bench("v2 RB=2 s=1 no loads/mfma/check", L2A(2, 1, 7));
  bench("v2 RB=4 s=2 no B loads", L2A(4, 2, 1));
  bench("v2 RB=4 s=2 no MFMA", L2A(4, 2, 2));
  bench("v2 RB=4 s=2 no event check", L2A(4, 2, 4));
  bench("v2 RB=4 s=2 no epilogue", L2A(4, 2, 8));
#define L7(W, SPL) [&] { hipLaunchKernelGGL((k_f16v7<W>), dim3(T / (8 * W), SPL), dim3(64 * W), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg, d_tile, d_info, chunks, SPL); }
#define L7A(W, SPL, AB) [&] { hipLaunchKernelGGL((k_f16v7<W, AB>), dim3(T / (8 * W), SPL), dim3(64 * W), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg, d_tile, d_info, chunks, SPL); }
#define L8(W, SPL) [&] { hipLaunchKernelGGL((k_f16v8<W>), dim3(T / (8 * W), SPL), dim3(64 * W), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part, d_dbg, d_tile, d_info, chunks, SPL); }
  bench("v8 4 waves splits=1", L8(4, 1));
  bench("v8 4 waves splits=2", L8(4, 2));
  bench("v8 8 waves splits=1", L8(8, 1));
  bench("v8 8 waves splits=2", L8(8, 2));
  bench("v8 8 waves splits=5", L8(8, 5));
  bench("v8 16 waves splits=1", L8(16, 1));
  bench("v8 16 waves splits=2", L8(16, 2));
  bench("v7 4w s=1 stage once", L7A(4, 1, 1));
  bench("v7 4w s=1 no event check", L7A(4, 1, 2));
  bench("v7 4w s=1 no recount code", L7A(4, 1, 4));
  bench("v7 4w s=1 all three", L7A(4, 1, 7));
  bench("v7 8w s=1 all three", L7A(8, 1, 7));
  bench("v7 4 waves splits=1", L7(4, 1));
  bench("v7 4 waves splits=2", L7(4, 2));
  bench("v7 4 waves splits=5", L7(4, 5));
  bench("v7 8 waves splits=1", L7(8, 1));
  bench("v7 8 waves splits=2", L7(8, 2));
  bench("v7 8 waves splits=5", L7(8, 5));
  bench("v7 16 waves splits=1", L7(16, 1));
  bench("v7 16 waves splits=2", L7(16, 2));
  bench("v2 RB=1 splits=1", L2(1, 1));
  bench("v2 RB=1 splits=2", L2(1, 2));
  bench("v2 RB=2 splits=1", L2(2, 1));
  bench("v2 RB=2 splits=2", L2(2, 2));
  bench("v2 RB=2 splits=5", L2(2, 5));
  bench("v2 RB=4 splits=2", L2(4, 2));
  bench("v2 RB=4 splits=5", L2(4, 5));
  bench("v2 RB=4 splits=10", L2(4, 10));
  return 0;
}
