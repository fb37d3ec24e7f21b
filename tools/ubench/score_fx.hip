// Prototype 2 for stage C2: TWO kernels.
//   F (filter): fp16-split residuals on the matrix pipe, definite inliers counted from sign bits, undecided tests
//               appended to a global queue, windows it cannot handle marked in a bitmap.  No fp32 re-evaluation inside.
//   X (exact):  the queued tests and the marked windows, canonical fp32, integer atomics into the counts.
//   bash tools/ubench/run.sh score_fx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float canon_d2(const float* M, const float* p) {
  const float ex = M[9] + fma_(M[2], p[2], fma_(M[1], p[1], fma_(M[0], p[0], -p[3])));
  const float ey = M[10] + fma_(M[5], p[2], fma_(M[4], p[1], fma_(M[3], p[0], -p[4])));
  const float ez = M[11] + fma_(M[8], p[2], fma_(M[7], p[1], fma_(M[6], p[0], -p[5])));
  return fma_(ez, ez, fma_(ey, ey, ex * ex));
}
constexpr int PC = 512;
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
    const float p[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    c0 += canon_d2(M, p) < tau2 ? 1u : 0u;
  }
  partial[(size_t)blockIdx.y * ldl + l] = c0;
}

// ------------------------------------------------------------------------------------------------------------
constexpr float RS = 1024.0f;
constexpr int UNIT = 256;            // points per staging unit (8 steps of 32)
constexpr int WIN = 1024;            // points per window (32 steps: one 32-bit shift register per test)
struct FilterInfo {                  // one per call, written by the tile kernel
  float s, pmax, qmax, pad;
};
constexpr int NQ = 256;             // sub-queues: a single ticket counter serialises ~6000 same-address atomics (~50 us)
struct FxCtl { uint32_t qcount[NQ][32]; uint32_t overflow; };  // one counter per 128-byte line

// maxima of |p| and |q| (order-free: atomicMax on the bit patterns of non-negative floats)
__global__ __launch_bounds__(256) void k_max(const float* __restrict__ planes, int n, int ld, uint32_t* __restrict__ mx) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  float mp = 0.f, mq = 0.f;
  if (m < n) {
    mp = fmaxf(fabsf(planes[m]), fmaxf(fabsf(planes[ld + m]), fabsf(planes[2 * ld + m])));
    mq = fmaxf(fabsf(planes[3 * (size_t)ld + m]), fmaxf(fabsf(planes[4 * (size_t)ld + m]), fabsf(planes[5 * (size_t)ld + m])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&mx[0], __float_as_uint(mp)); atomicMax(&mx[1], __float_as_uint(mq)); }
}
// the fp16 tile: 32 B per point, k order [Pxh Pxl Pxh Pyh Pyl Pyh Pzh Pzl | Pzh Qxh Qxl Qyh Qyl Qzh Qzl 0]; rows [n, rows) are
// sentinels (far away).  One scale for the whole call.
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ planes, int n, int ld, int rows, const uint32_t* __restrict__ mx,
                                              uint4* __restrict__ tile, FilterInfo* __restrict__ info) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const float Pmax = __uint_as_float(mx[0]), Qmax = __uint_as_float(mx[1]), mxv = fmaxf(Pmax, Qmax);
  int e = (int)((__float_as_uint(mxv) >> 23) & 255u) - 127;
  int k = 8 - e; k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float s = __uint_as_float((uint32_t)(k + 127) << 23);
  if (m == 0) *info = FilterInfo{s, Pmax, Qmax, 0.f};
  if (m >= rows) return;
  _Float16 hi[6], lo[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const float X = m < n ? planes[(size_t)c * ld + m] * s : (c < 3 ? 0.f : 32768.f);
    hi[c] = (_Float16)X; lo[c] = (_Float16)(X - (float)hi[c]);
  }
  half8 f0 = {hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2]};
  half8 f1 = {hi[2], hi[3], lo[3], hi[4], lo[4], hi[5], lo[5], (_Float16)0.f};
  tile[(size_t)m * 2] = *reinterpret_cast<uint4*>(&f0); tile[(size_t)m * 2 + 1] = *reinterpret_cast<uint4*>(&f1);
}

constexpr int QL = 256;  // LDS queue entries per wave
// F.  Workgroup = WAVES waves, a wave = 8 hypotheses (one 32-row block); grid.y = splits of the windows.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 6) void k_filter(const float* __restrict__ Rt, uint32_t ldl, float tau2,
                                                          const uint4* __restrict__ tile, const FilterInfo* __restrict__ info,
                                                          int windows, int splits, uint32_t* __restrict__ cnt_out,
                                                          uint2* __restrict__ gq, uint32_t gq_cap, FxCtl* __restrict__ ctl,
                                                          uint32_t* __restrict__ bitmap) {
  __shared__ uint4 Bt[2][UNIT * 2];
  __shared__ float4 Ttab[WAVES][8];
  __shared__ uint32_t queue[WAVES][QL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hf = lane >> 5;
  const int per = (windows + splits - 1) / splits, w0 = blockIdx.y * per, w1 = min(windows, w0 + per);
  const uint32_t wid = blockIdx.x * WAVES + wave, h0 = wid * 8;
  const int u0 = w0 * (WIN / UNIT), u1 = w1 * (WIN / UNIT);  // staging units of this block
  // 8 KiB per unit: 512 x 16 bytes by LDS-DMA.  Issued through asm so that hipcc does not count it: the builtin makes
  // the compiler wait vmcnt(0) before the NEXT ds_read (it cannot tell the two buffers apart), which exposes the whole
  // DMA latency in every unit.  Our own wait sits before the barrier that publishes the buffer.
  auto stage = [&](int u, int buf) {
#pragma unroll
    for (int i = 0; i < 8 / WAVES; i++) {
      const uint4* gsrc = tile + (size_t)u * (UNIT * 2) + 64 * WAVES * i + tid;
      const uint32_t lds_dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&Bt[buf][64 * WAVES * i + wave * 64]));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  if (u0 < u1) stage(u0, 0);
  const FilterInfo fi = *info;
  half8 A;
  {
    const int r = lane & 31, hy = r >> 2, c = r & 3;
    float x[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) x[kk] = Rt[(size_t)(3 * (c < 3 ? c : 0) + kk) * ldl + h0 + hy];  // unconditional: one round trip
#pragma unroll
    for (int kk = 0; kk < 3; kk++) x[kk] = c < 3 ? x[kk] * RS : 0.f;
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { rh[kk] = (_Float16)x[kk]; rl[kk] = (_Float16)(x[kk] - (float)rh[kk]); }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A = hf ? a1 : a0;
  }
  float tmax = 0.f; bool wild = false;
  if (lane < 8) {
    const uint32_t h = h0 + lane;
    float v[12];
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = Rt[(size_t)c * ldl + h];  // all twelve in flight together (no short-circuit between them)
    float rmax = 0.f;
#pragma unroll
    for (int c = 0; c < 9; c++) rmax = fmaxf(rmax, fabsf(v[c]));
    tmax = fmaxf(fabsf(v[9]), fmaxf(fabsf(v[10]), fabsf(v[11])));
    float sum = 0.f;  // NaN anywhere poisons the sum (fmaxf drops NaNs)
#pragma unroll
    for (int c = 0; c < 12; c++) sum += v[c] * 0.f;
    wild = !(rmax <= 1.5f) || !(tmax < 1e30f) || !(sum == 0.f);
    Ttab[wave][lane] = make_float4(v[9] * RS * fi.s, v[10] * RS * fi.s, v[11] * RS * fi.s, 0.f);
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  tmax = __shfl(tmax, 0);
  const bool any_wild = __ballot(wild) != 0;
  const float s = fi.s, st = s * sqrtf(tau2);
  const float Sb = 2.6f * (fi.pmax * s) + fi.qmax * s + tmax * s;
  const float eta = Sb * (1.0f / 65536.0f);
  const bool fast = !any_wild && (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f);  // false on NaN
  const float lo_e = RS * (st - eta), hi_e = RS * (st + eta);
  const float LO = lo_e * lo_e * (1.0f - 1e-6f), HI = hi_e * hi_e * (1.0f + 1e-6f);
  const uint32_t W2b = fast ? __float_as_uint(HI - LO) : 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  f32x16 C;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const float4 T = Ttab[wave][2 * jj + hf];
    C[4 * jj] = T.x; C[4 * jj + 1] = T.y; C[4 * jj + 2] = T.z; C[4 * jj + 3] = 0.f;
  }
  uint32_t total[4] = {0, 0, 0, 0};
  uint32_t sr[4] = {0, 0, 0, 0};
  uint32_t qn = 0, qwin = 0;  // queue fill; fill at the start of the current window
  bool over = false;
  uint32_t* q = queue[wave];
  auto flush = [&]() {  // the wave's queue -> the global queue (one ticket)
    uint32_t base = 0;
    const uint32_t sq = wid % NQ, cap = gq_cap / NQ;
    if (lane == 0) base = atomicAdd(&ctl->qcount[sq][0], qn);
    base = __shfl(base, 0);
    if (base + qn <= cap) {
      for (uint32_t i = lane; i < qn; i += 64) gq[(size_t)sq * cap + base + i] = make_uint2(q[i] >> 5, (wid << 5) | (q[i] & 31u));
    } else if (lane == 0) ctl->overflow = 1u;
    qn = 0;
  };
  for (int u = u0; u < u1; u++) {
    const int buf = (u - u0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of unit u has landed
    __syncthreads();
    if (u + 1 < u1) stage(u + 1, buf ^ 1);
    if (fast) {
      const half8* Bc = reinterpret_cast<const half8*>(Bt[buf]) + col * 2 + hf;
      half8 b = Bc[0];
#pragma unroll 2
      for (int g = 0; g < UNIT / 32; g++) {
        const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
        if (g + 1 < UNIT / 32) b = Bc[64 * (g + 1)];
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
          if (qn + k2 <= QL) {
            uint32_t bits = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) bits |= (__float_as_uint(x[t]) < W2b) ? (1u << t) : 0u;
            if (mn < W2b) q[qn + __popcll(hm & ((1ull << lane) - 1ull))] = ((uint32_t)(u * UNIT + 32 * g + col) << 5) | ((uint32_t)hf << 4) | bits;
            qn += k2;
          } else over = true;
        }
      }
    }
    if ((u + 1) % (WIN / UNIT) == 0) {  // window boundary
      if (!fast || over) {
        if (lane == 0) atomicOr(&bitmap[((size_t)wid * windows + u / (WIN / UNIT)) >> 5], 1u << (((size_t)wid * windows + u / (WIN / UNIT)) & 31));
        qn = qwin;  // entries of a recounted window are dropped
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++) total[t] += (uint32_t)__popc(sr[t]);
      }
#pragma unroll
      for (int t = 0; t < 4; t++) sr[t] = 0;
      over = false;
      if (qn > QL / 2) flush();
      qwin = qn;
    }
  }
  if (qn) flush();
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t c = total[t];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
    if (col == 0) cnt_out[(size_t)blockIdx.y * ldl + h0 + 2 * t + hf] = c;
  }
}

// X: queue entries (one per thread, grid stride), then marked windows (one per workgroup, grid stride over bitmap bits)
__global__ __launch_bounds__(256) void k_exact(const float* __restrict__ planes, int n, int ld, const float* __restrict__ Rt,
                                               uint32_t ldl, float tau2, int windows, uint32_t n_waves,
                                               const uint2* __restrict__ gq, uint32_t gq_cap, const FxCtl* __restrict__ ctl,
                                               const uint32_t* __restrict__ bitmap, uint32_t* __restrict__ cnt_out) {
  const uint32_t sq = blockIdx.x % NQ, cap = gq_cap / NQ, nq = min(ctl->qcount[sq][0], cap);
  for (uint32_t i = (blockIdx.x / NQ) * 256 + threadIdx.x; i < nq; i += (gridDim.x / NQ) * 256) {
    const uint2 e = gq[(size_t)sq * cap + i];
    const uint32_t m = e.x, wid = e.y >> 5, ehf = (e.y >> 4) & 1;
    const size_t bit = (size_t)wid * windows + m / WIN;
    if ((bitmap[bit >> 5] >> (bit & 31)) & 1u) continue;  // the window is recounted as a whole
    float p[6];
#pragma unroll
    for (int c = 0; c < 6; c++) p[c] = planes[(size_t)c * ld + m];
    for (uint32_t bits = e.y & 0xFu; bits; bits &= bits - 1) {
      const uint32_t h = wid * 8 + 2 * (__ffs(bits) - 1) + ehf;
      float M[12];
#pragma unroll
      for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + h];
      if (canon_d2(M, p) < tau2) atomicAdd(&cnt_out[h], 1u);
    }
  }
  const size_t nbits = (size_t)n_waves * windows;
  for (size_t w = blockIdx.x; w < (nbits + 31) / 32; w += gridDim.x) {
    uint32_t word = bitmap[w];
    while (word) {
      const size_t bit = w * 32 + (__ffs(word) - 1); word &= word - 1;
      const uint32_t wid = (uint32_t)(bit / windows), win = (uint32_t)(bit % windows);
      const uint32_t h = wid * 8 + (threadIdx.x & 7);
      float M[12];
#pragma unroll
      for (int c = 0; c < 12; c++) M[c] = Rt[(size_t)c * ldl + h];
      uint32_t cnt = 0;
      for (int i = threadIdx.x >> 3; i < WIN; i += 32) {
        const int m = win * WIN + i;
        if (m < n) {
          float p[6];
#pragma unroll
          for (int c = 0; c < 6; c++) p[c] = planes[(size_t)c * ld + m];
          cnt += canon_d2(M, p) < tau2 ? 1u : 0u;
        }
      }
      cnt += __shfl_xor(cnt, 8); cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 32);
      if ((threadIdx.x & 63) < 8 && cnt) atomicAdd(&cnt_out[h], cnt);
    }
  }
}

static uint64_t checksum(const std::vector<uint32_t>& p, uint32_t ldl, int rows, uint32_t T) {
  uint64_t s = 0;
  for (uint32_t h = 0; h < T; h++) { uint64_t c = 0; for (int k = 0; k < rows; k++) c += p[(size_t)k * ldl + h]; s = s * 1000003ull + c; }
  return s;
}

int main(int argc, char** argv) {
  const float L = argc > 1 ? atof(argv[1]) : 3.0f, tau = argc > 2 ? atof(argv[2]) : 0.1f;
  const int n = argc > 3 ? atoi(argv[3]) : 5000; const uint32_t T = argc > 4 ? atoi(argv[4]) : 50176;
  const int nwild = argc > 5 ? atoi(argv[5]) : 0;
  const int ld = (n + 1023) / 1024 * 1024; const uint32_t ldl = T; const int chunks = (n + PC - 1) / PC;
  const int windows = (n + WIN - 1) / WIN, rows = windows * WIN;
  std::vector<float> planes(6 * (size_t)ld, 0.f), Rt(12 * (size_t)ldl);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  const float ang = 0.7f, cg = cosf(ang), sg = sinf(ang), tg[3] = {0.3f * L, -0.2f * L, 0.1f * L};
  for (int m = 0; m < n; m++) {
    float p[3] = {rnd() * L / 2, rnd() * L / 2, rnd() * L / 2};
    float q[3] = {cg * p[0] - sg * p[1] + tg[0], sg * p[0] + cg * p[1] + tg[1], p[2] + tg[2]};
    if (m % 7 != 0) for (int c = 0; c < 3; c++) q[c] = rnd() * L; else for (int c = 0; c < 3; c++) q[c] += tau / 3 * rnd();
    for (int c = 0; c < 3; c++) { planes[c * ld + m] = p[c]; planes[(3 + c) * ld + m] = q[c]; }
  }
  for (uint32_t h = 0; h < T; h++) {
    const float a = ang + 0.02f * rnd(), ch = cosf(a), sh = sinf(a);
    const float R[9] = {ch, -sh, 0, sh, ch, 0, 0, 0, 1};
    for (int c = 0; c < 9; c++) Rt[(size_t)c * ldl + h] = R[c];
    for (int c = 0; c < 3; c++) Rt[(size_t)(9 + c) * ldl + h] = tg[c] + 0.5f * tau * rnd();
  }
  for (int i = 0; i < nwild; i++) {  // wild hypotheses: scaled rotation, huge translation, a NaN
    const uint32_t h = (uint32_t)rand() % T;
    if (i % 3 == 0) for (int c = 0; c < 9; c++) Rt[(size_t)c * ldl + h] *= 3.f;
    else if (i % 3 == 1) Rt[(size_t)9 * ldl + h] = 1e6f;
    else Rt[(size_t)4 * ldl + h] = NAN;
  }
  float *d_pl, *d_Rt; uint32_t *d_part;
  hipMalloc(&d_pl, planes.size() * 4); hipMalloc(&d_Rt, Rt.size() * 4);
  const int prow = chunks > 16 ? chunks : 16;
  hipMalloc(&d_part, (size_t)prow * ldl * 4);
  hipMemcpy(d_pl, planes.data(), planes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_Rt, Rt.data(), Rt.size() * 4, hipMemcpyHostToDevice);
  const float tau2 = tau * tau;
  std::vector<uint32_t> part((size_t)prow * ldl);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t* d_mx; uint4* d_tile; FilterInfo* d_info; uint2* d_gq; FxCtl* d_ctl; uint32_t* d_bitmap;
  const uint32_t n_waves = T / 8, gq_cap = 1u << 20;
  const size_t bm_words = ((size_t)n_waves * windows + 31) / 32;
  hipMalloc(&d_mx, 8); hipMalloc(&d_tile, (size_t)(rows + UNIT) * 32); hipMalloc(&d_info, sizeof(FilterInfo));
  hipMalloc(&d_gq, (size_t)gq_cap * 8); hipMalloc(&d_ctl, sizeof(FxCtl)); hipMalloc(&d_bitmap, bm_words * 4);
  const char* only = argc > 6 ? argv[6] : nullptr;
  auto bench = [&](const char* name, int rows_out, auto launch) {
    if (only && !strstr(name, only)) return;
    hipMemset(d_part, 0, part.size() * 4);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(part.data(), d_part, part.size() * 4, hipMemcpyDeviceToHost);
    FxCtl c; hipMemcpy(&c, d_ctl, sizeof(c), hipMemcpyDeviceToHost);
    std::vector<uint32_t> bm(bm_words); hipMemcpy(bm.data(), d_bitmap, bm_words * 4, hipMemcpyDeviceToHost);
    size_t marked = 0; for (uint32_t w : bm) marked += __builtin_popcount(w);
    uint64_t tot = 0; for (uint32_t h = 0; h < T; h++) for (int k = 0; k < rows_out; k++) tot += part[(size_t)k * ldl + h];
    uint32_t qtot = 0, qmax = 0; for (int i = 0; i < NQ; i++) { qtot += c.qcount[i][0]; qmax = qmax > c.qcount[i][0] ? qmax : c.qcount[i][0]; }
    printf("%-30s %8.1f us  checksum %016llx  mean inliers %.1f  queued %u (max per sub-queue %u, overflow %u)  windows marked %zu\n", name, ms * 1000 / 20,
           (unsigned long long)checksum(part, ldl, rows_out, T), (double)tot / T, qtot, qmax, c.overflow, marked);
  };
  printf("L = %g tau = %g n = %d T = %u wild = %d\n", L, tau, n, T, nwild);
  bench("valu fp32 (lane = hypothesis)", chunks, [&] { hipLaunchKernelGGL(k_valu, dim3(T / 256, chunks), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, d_part); });
  auto prep = [&] {
    hipMemsetAsync(d_mx, 0, 8, 0);
    hipLaunchKernelGGL(k_max, dim3((n + 255) / 256), dim3(256), 0, 0, d_pl, n, ld, d_mx);
    hipLaunchKernelGGL(k_tile, dim3((rows + UNIT + 255) / 256), dim3(256), 0, 0, d_pl, n, ld, rows + UNIT, d_mx, d_tile, d_info);
  };
  prep();
  bench("prep (max + tile)", 0, prep);
#define FX(W, SPL, WITHX) [&] { \
    hipMemsetAsync(d_ctl, 0, sizeof(FxCtl), 0); hipMemsetAsync(d_bitmap, 0, bm_words * 4, 0); \
    hipLaunchKernelGGL((k_filter<W>), dim3(T / (8 * W), SPL), dim3(64 * W), 0, 0, d_Rt, ldl, tau2, d_tile, d_info, windows, SPL, d_part, d_gq, gq_cap, d_ctl, d_bitmap); \
    if (WITHX) hipLaunchKernelGGL(k_exact, dim3(512), dim3(256), 0, 0, d_pl, n, ld, d_Rt, ldl, tau2, windows, n_waves, d_gq, gq_cap, d_ctl, d_bitmap, d_part); }
  const int sp[] = {1, 2, 3, 5};
  for (int spl : sp) {
    if (spl > windows) break;
    char nm[64];
    snprintf(nm, 64, "F only  4 waves splits=%d", spl); bench(nm, spl, FX(4, spl, 0));
    snprintf(nm, 64, "F + X   4 waves splits=%d", spl); bench(nm, spl, FX(4, spl, 1));
    snprintf(nm, 64, "F + X   8 waves splits=%d", spl); bench(nm, spl, FX(8, spl, 1));
  }
  return 0;
}
