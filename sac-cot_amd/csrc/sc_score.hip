// sc_score.hip — stage C: per-triangle rigid transform (C1), hypothesis x correspondence inlier counting
// (C2) and the winner's mask (C3).  SURVEY.md §8a rows C1, C2, C3.
//
// C2 is the arithmetic-heavy kernel of the path (T*N tests, 27 flop each) and moves almost no HBM bytes
// (48 B per hypothesis in, 4 B out), so it is bounded by the fp32 vector rate, not by HBM.  Mapping:
//   lane          = one hypothesis: its 12 coefficients live in VGPRs for the whole kernel, its inlier
//                   count is a private register — no cross-lane reduction, no atomics on the hot loop;
//   workgroup     = 256 hypotheses x one chunk of <= 1024 correspondences staged once into LDS;
//   inner loop    = every lane reads the SAME point (one ds_read_b128 + one ds_read_b64, LDS broadcast,
//                   conflict-free), then 12 FMA-class ops for the residual, 3 for its square norm, one
//                   compare and one add-with-carry: 17 VALU instructions per test;
//   grid          = ceil(T/256) x chunks, so a 50k x 5k problem is ~1000 workgroups (~4 waves per SIMD);
//   partial counts are stored coalesced ([chunk][hypothesis]) and summed by the arg-max kernel.
#include <cstddef>
#include <vector>
#include <hip/hip_ext.h>

#include "sc_arith.hpp"
#include "sc_block.hpp"
#include "sc_kernels.hpp"
#include "sc_gramref.hpp"

namespace sc {

// ------------------------------------------------------------------------------------------------
// sharding of the ranked list: blocks of `block` triangles dealt round-robin to ranks
// ------------------------------------------------------------------------------------------------

uint32_t shard_local_count(uint32_t T_eff, uint32_t block, uint32_t rank, uint32_t world) {
  uint64_t n = 0;
  for (uint64_t gb = rank; gb * block < T_eff; gb += world) {
    const uint64_t lo = gb * block, hi = (gb + 1) * (uint64_t)block;
    n += (hi < T_eff ? hi : T_eff) - lo;
  }
  return (uint32_t)n;
}

// ---- Gram filter: constants shared by host and device -------------------------------------------------------------
constexpr int GX_UNIT = 256;     // correspondences per staging unit (8 MFMA steps of 32): 16 KiB; LDS holds three
constexpr int GX_TILE_B = 64;    // bytes of the tile per correspondence: hi halves and lo halves of its 16 features
constexpr int GX_TILE_Q = GX_TILE_B / 16;  // ... in 16-byte pieces
constexpr int GX_WAVES = 8;      // waves per workgroup, 32 hypotheses each: eight waves share a tile, so each of them issues three 1 KiB
                                 // LDS-DMA pieces per eight steps (with 4 waves and 128-correspondence units the DMA issue alone cost a quarter of the kernel)
constexpr int GX_QL = 128;       // LDS queue entries per wave (8 bytes each)
constexpr float GX_RS = 256.0f;  // scale of the A operand (keeps the low halves of the coefficients out of fp16's sub-normal range)
constexpr double GX_ACC = 1.1e-6;    // 18.5 x 2^-24: error of one MFMA per unit of its LARGEST term (five times the largest seen: score_gram_kernel)
constexpr double GX_Q = 7.5e-7;      // 3.01 x 2^-22 (+ margin): the dropped lo x lo products and split remainders per unit of sum |w F|
constexpr double GX_NORM = 2.5e-7;   // 2^-22 (+ margin): what the norm feature's two fp16 pieces leave, per unit of max |V'|^2
constexpr double GX_CANON = 4.2e-7;  // sqrt(3) * 4 * 2^-24: deviation of the canonical fp32 residual VECTOR per unit of magnitude
// shell half-width of a hypothesis, in units of the scaled squared residual.  F = |dM|_F, m = max |dM_ij|, Tn = |tau'|, g = the defect of
// R^T R; the correspondences it is tested against have |V'| <= vb and |Q'| <= Qb.  Mh: the largest single term of the 48-term
// dot product; Sl: the sum of the absolute values of the 15 split terms; S: the sum of all |terms|.  (Derivation: score_gram_kernel.)
__host__ __device__ inline double gram_eps(double F, double m, double Tn, double g, double Pn, double Qb, double vb, double st) {
  const double t1 = 2.0 * (m + 1.5 * g) * Qb * Pn, t2 = vb * vb, t3 = 2.0 * F * Tn * Pn, t4 = 2.0 * Tn * vb, t5 = Tn * Tn + 1.5 * st * st;
  double Mh = t1;
  Mh = t2 > Mh ? t2 : Mh; Mh = t3 > Mh ? t3 : Mh; Mh = t4 > Mh ? t4 : Mh; Mh = t5 > Mh ? t5 : Mh;
  Mh *= 1.05;
  const double Sl = 2.0 * (F + 4.5 * g) * Qb * Pn + t3 + t4, S = Sl + t2 + t5;
  return GX_ACC * Mh + GX_Q * Sl + GX_NORM * t2 + 1e-8 * S + 3.0 * g * vb * Pn + 1e-4;
}
// the same with the SUM of the terms in place of their maximum: an upper bound of gram_eps that is a quadratic in vb with positive
// coefficients (gram_coef_block: a shell made for |V'| <= vb_h is safe beyond vb_h if this stays under 0.1 tau'^2)
__host__ __device__ inline double gram_eps_sum(double F, double m, double Tn, double g, double Pn, double Qb, double vb, double st) {
  const double t1 = 2.0 * (m + 1.5 * g) * Qb * Pn, t2 = vb * vb, t3 = 2.0 * F * Tn * Pn, t4 = 2.0 * Tn * vb, t5 = Tn * Tn + 1.5 * st * st;
  const double Sl = 2.0 * (F + 4.5 * g) * Qb * Pn + t3 + t4, S = Sl + t2 + t5;
  return GX_ACC * 1.05 * (t1 + t2 + t3 + t4 + t5) + GX_Q * Sl + GX_NORM * t2 + 1e-8 * S + 3.0 * g * vb * Pn + 1e-4;
}

// ------------------------------------------------------------------------------------------------
// C1
// ------------------------------------------------------------------------------------------------
__device__ void filter_tile_block(const float* __restrict__ planes, int n, int ld, const FilterTileJob& job, uint32_t block,
                                  uint32_t blocks);
__device__ void gram_coef_block(const GramCoef& coef, const float v[12], uint32_t l, uint32_t ldl, uint32_t n_local, float tau2);

__global__ __launch_bounds__(256) void kabsch_shard_kernel(const float* __restrict__ planes, int n, int ld, TriSource ts,
                                                           Shard sh, float* __restrict__ RtSoA,
                                                           float* __restrict__ RtAoS, FilterTileJob job,
                                                           uint32_t kabsch_blocks,
                                                           const uint64_t* __restrict__ t_eff_dev) {
  if (blockIdx.x >= kabsch_blocks) {  // the extra workgroups: C2's fp16 tile of the correspondences (see filter_tile_block)
    filter_tile_block(planes, n, ld, job, blockIdx.x - kabsch_blocks, gridDim.x - kabsch_blocks);
    return;
  }
  const uint32_t l = blockIdx.x * 256 + threadIdx.x;  // < ld_local: a multiple of 256, kabsch_blocks = ld_local / 256
  float Rt[12];
  // t_eff_dev: the launch was sized for sh.T_eff = the requested T before the host knew how many triangles there are; a
  // position beyond the real selection holds nothing that may be dereferenced (the host repeats such a call: sc_capi.hip)
  const uint32_t g = l < sh.n_local ? shard_global_index(l, sh.block, sh.rank, sh.world) : 0u;
  uint32_t v[3];
  if (l < sh.n_local && (!t_eff_dev || (uint64_t)g < *t_eff_dev) && tri_lookup(ts, g, v)) {
    float P[9], Q[9];
    load_triangle(planes, ld, v, P, Q);
    kabsch3(P, Q, Rt);
  } else {
#pragma unroll
    for (int c = 0; c < 12; c++) Rt[c] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 12; c++) RtSoA[(size_t)c * sh.ld_local + l] = Rt[c];
  if (RtAoS) {  // 12 consecutive floats per hypothesis: what the lane = correspondence scoring kernel loads as scalars
    float4* o = reinterpret_cast<float4*>(RtAoS + 12 * (size_t)l);
    o[0] = make_float4(Rt[0], Rt[1], Rt[2], Rt[3]);
    o[1] = make_float4(Rt[4], Rt[5], Rt[6], Rt[7]);
    o[2] = make_float4(Rt[8], Rt[9], Rt[10], Rt[11]);
  }
  // the Gram filter's coefficients of this hypothesis (whole workgroups get here: ld_local is a multiple of 256)
  if (job.mode == 2) gram_coef_block(job.coef, Rt, l, sh.ld_local, sh.n_local, job.tau2);
}

void launch_kabsch(const Points& pts, const TriSource& ts, const Shard& sh, float* RtSoA, float* RtAoS,
                   const FilterTileJob* tile_job, hipStream_t st, const uint64_t* t_eff_dev) {
  if (sh.ld_local == 0) return;
  const uint32_t kb = sh.ld_local / 256, tb = tile_job ? (tile_job->rows + 255) / 256 : 0u;
  hipLaunchKernelGGL(kabsch_shard_kernel, dim3(kb + tb), dim3(256), 0, st, pts.planes, pts.n, pts.ld, ts, sh, RtSoA, RtAoS,
                     tile_job ? *tile_job : FilterTileJob{}, kb, t_eff_dev);
}

__global__ __launch_bounds__(256) void kabsch_aos_kernel(const float* __restrict__ planes, int ld,
                                                         const uint32_t* __restrict__ tri, uint32_t T,
                                                         float* __restrict__ Rt_out) {
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  if (h >= T) return;
  float P[9], Q[9], Rt[12];
  load_triangle(planes, ld, tri + 3 * (size_t)h, P, Q);
  kabsch3(P, Q, Rt);
#pragma unroll
  for (int c = 0; c < 12; c++) Rt_out[12 * (size_t)h + c] = Rt[c];
}

void launch_kabsch_aos(const Points& pts, const uint32_t* tri, uint32_t T, float* Rt, hipStream_t st) {
  if (T == 0) return;
  hipLaunchKernelGGL(kabsch_aos_kernel, dim3((T + 255) / 256), dim3(256), 0, st, pts.planes, pts.ld, tri, T, Rt);
}

__global__ __launch_bounds__(256) void rt_to_soa_kernel(const float* __restrict__ Rt, uint32_t T, uint32_t ld_local,
                                                        float* __restrict__ RtSoA) {
  const uint32_t l = blockIdx.x * 256 + threadIdx.x;
  if (l >= ld_local) return;
#pragma unroll
  for (int c = 0; c < 12; c++) RtSoA[(size_t)c * ld_local + l] = (l < T) ? Rt[12 * (size_t)l + c] : 0.0f;
}

void launch_rt_to_soa(const float* Rt, uint32_t T, uint32_t ld_local, float* RtSoA, hipStream_t st) {
  if (ld_local == 0) return;
  hipLaunchKernelGGL(rt_to_soa_kernel, dim3(ld_local / 256), dim3(256), 0, st, Rt, T, ld_local, RtSoA);
}

// ------------------------------------------------------------------------------------------------
// C2
// ------------------------------------------------------------------------------------------------
constexpr int SCORE_THREADS = 256;
constexpr int SCORE_PC = 512;  // correspondences per chunk (LDS: 12 KiB per workgroup, 8 workgroups per CU)

// Point chunking of C2: workgroups = (hypothesis groups of 256) x chunks.  Chunks are as long as LDS allows (512) for
// big problems, shorter for small ones so that the launch still has ~2048 workgroups (8 per CU) to fill the chip.
void score_plan(int n, uint32_t ld_local, uint32_t* chunks, int* chunk_pts) {
  const uint32_t groups = ld_local / SCORE_THREADS ? ld_local / SCORE_THREADS : 1u;
  uint32_t want = 2048 / groups;                                      // at most one resident generation (8 per CU)
  const uint32_t min_chunks = (uint32_t)((n + SCORE_PC - 1) / SCORE_PC);  // LDS capacity
  const uint32_t max_chunks = (uint32_t)((n + 63) / 64);                  // at least 64 points per chunk
  if (want < min_chunks) want = min_chunks;
  if (want > max_chunks) want = max_chunks;
  if (want < 1) want = 1;
  int per = (int)((n + want - 1) / want);
  per = (per + 3) & ~3;                                               // multiple of 4, <= SCORE_PC
  if (per > SCORE_PC) per = SCORE_PC;
  *chunk_pts = per;
  *chunks = (uint32_t)((n + per - 1) / per);
}
// (r02 also had a lane = correspondence mapping with the hypothesis in scalar registers: bit-exact, 114.8 us against 93.9 at C2
// — VALU instructions with a scalar-register operand issue at 4.4 cycles on this part against 3.04 — removed in r05;
// profiles/r02_* keep its numbers.)
uint32_t score_chunks(int n, uint32_t ld_local) {
  uint32_t c; int p;
  score_plan(n, ld_local, &c, &p);
  return c;
}

// One inlier test, written so that hipcc keeps 17 single-issue VALU ops (build.py passes -fno-slp-vectorize:
// SLP packing turns the chain into v_pk_* plus ~5 v_mov per test, which is slower on gfx950).
// MODE 0: 1 if inlier.  MODE 1 / 2 (include/saccot.h, SC_SCORE_MSE / _MAE): floor(1024 max(0, 1 - d2 / tau^2)) resp.
// floor(1024 max(0, 1 - d / tau)) — integers, so the sum over the correspondences is exact in any order.  `thr` is
// tau^2, 1 / tau^2 or 1 / tau.  (d2 = +inf -> fma = -inf -> 0; the float -> u32 conversion truncates.)
template <int MODE>
__device__ __forceinline__ uint32_t inlier_bit(const float (&M)[12], const float4 a, const float2 b, float thr) {
  const float d2 = resid2(M, a.x, a.y, a.z, a.w, b.x, b.y);
  if (MODE == 0) return d2 < thr ? 1u : 0u;
  const float x = MODE == 1 ? d2 : sqrt_rn(d2);
  return (uint32_t)(fmaxf(fma_(-x, thr, 1.0f), 0.0f) * 1024.0f);
}

// VALU body: lane = hypothesis.  bx = workgroup index along the hypotheses (256 per workgroup), hyp_base = first one.
template <int MODE>
__device__ __forceinline__ void score_valu_body(float4* __restrict__ smem, uint32_t hyp_base, int chunk,
                                                const float* __restrict__ planes, int n, int ld,
                                                const float* __restrict__ RtSoA, uint32_t ld_local, float tau2,
                                                int chunk_pts, uint32_t* __restrict__ partial) {
  float4* pA = smem;                                          // px py pz qx
  float2* pB = reinterpret_cast<float2*>(smem + SCORE_PC);    // qy qz
  const int m0 = chunk * chunk_pts;
  const int cnt_pts = min(chunk_pts, n - m0);
  const int padded = (cnt_pts + 3) & ~3;  // <= chunk_pts <= SCORE_PC (chunk_pts is a multiple of 4)
  for (int t = threadIdx.x; t < padded; t += SCORE_THREADS) {
    const int m = m0 + t;
    if (t < cnt_pts) {
      pA[t] = make_float4(planes[m], planes[(size_t)ld + m], planes[2 * (size_t)ld + m], planes[3 * (size_t)ld + m]);
      pB[t] = make_float2(planes[4 * (size_t)ld + m], planes[5 * (size_t)ld + m]);
    } else {  // sentinel: p = 0, q = 1e30 -> residual ~1e30, squared = +inf: never < tau2, score 0, never NaN
      pA[t] = make_float4(0.f, 0.f, 0.f, 1e30f);
      pB[t] = make_float2(1e30f, 1e30f);
    }
  }
  const uint32_t l = hyp_base + threadIdx.x;
  float M[12];
#pragma unroll
  for (int c = 0; c < 12; c++) M[c] = RtSoA[(size_t)c * ld_local + l];
  const bool ok = finite12(M);
  __syncthreads();
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (int t = 0; t < padded; t += 4) {
    const float4 a0 = pA[t], a1 = pA[t + 1], a2 = pA[t + 2], a3 = pA[t + 3];
    const float2 b0 = pB[t], b1 = pB[t + 1], b2 = pB[t + 2], b3 = pB[t + 3];
    c0 += inlier_bit<MODE>(M, a0, b0, tau2);
    c1 += inlier_bit<MODE>(M, a1, b1, tau2);
    c2 += inlier_bit<MODE>(M, a2, b2, tau2);
    c3 += inlier_bit<MODE>(M, a3, b3, tau2);
  }
  partial[(size_t)chunk * ld_local + l] = ok ? (c0 + c1) + (c2 + c3) : 0u;
}

// ------------------------------------------------------------------------------------------------
// C2 on the matrix pipe.  v_mfma_f32_16x16x4_f32 computes D = A(16x4) B(4x16) + C as a k-ordered fp32 fma chain
// (one rounding per product, bit-for-bit fmaf: cdna guide §3 "FP32-input MFMA"), which is exactly the canonical
// residual  e_c = fma(t_c,1, fma(r_c2,pz, fma(r_c1,py, fma(r_c0,px, -q_c))))  when
//   A[row = 4 hq + c][k] = (r_c0, r_c1, r_c2, t_c)[k]      4 hypotheses hq x 3 components c (row 4hq+3 is zero)
//   B[k][col]            = (px, py, pz, 1)[k] of point col   16 correspondences
//   C[row][col]          = -q_c of point col
// so one instruction yields the residuals of 4 x 16 tests.  Lane l receives D rows 4 (l>>4) + {0,1,2} of column
// l & 15: the three components of ONE test, so the squared norm, the compare and the count stay lane-local (5 VALU
// instructions per 64 tests instead of 17.5), and 4 shuffles per hypothesis at the very end sum the 16 columns.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MF_QUADS = 8;                 // hypothesis quads per wave: 32 hypotheses
constexpr int MF_HYPS_PER_BLOCK = 4 * 4 * MF_QUADS;  // 4 waves

// MFMA body: hyp_base = first hypothesis of the workgroup (MF_HYPS_PER_BLOCK per workgroup).
__device__ __forceinline__ void score_mfma_body(float4* __restrict__ smem, uint32_t hyp_base, int chunk,
                                                const float* __restrict__ planes, int n, int ld,
                                                const float* __restrict__ RtSoA, uint32_t ld_local, float tau2,
                                                int chunk_pts, uint32_t* __restrict__ partial) {
  float4* P4 = smem;             // px py pz 1
  float4* Qn = smem + SCORE_PC;  // -qx -qy -qz 0
  const int m0 = chunk * chunk_pts;
  const int cnt_pts = min(chunk_pts, n - m0);
  const int padded = (cnt_pts + 15) & ~15;  // whole 16-point groups; chunk_pts <= SCORE_PC and SCORE_PC % 16 == 0
  for (int t = threadIdx.x; t < padded; t += 256) {
    const int m = m0 + t;
    if (t < cnt_pts) {
      P4[t] = make_float4(planes[m], planes[(size_t)ld + m], planes[2 * (size_t)ld + m], 1.0f);
      Qn[t] = make_float4(-planes[3 * (size_t)ld + m], -planes[4 * (size_t)ld + m], -planes[5 * (size_t)ld + m], 0.f);
    } else {  // sentinel: residual ~ -1e30, squared = +inf, never < tau2, never NaN
      P4[t] = make_float4(0.f, 0.f, 0.f, 1.0f);
      Qn[t] = make_float4(-1e30f, -1e30f, -1e30f, 0.f);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t hyp0 = hyp_base + wave * (4 * MF_QUADS);  // first hypothesis of this wave
  // A operands: lane l feeds row (l & 15) = 4 hq + c, k = l >> 4
  const int a_hq = (lane & 15) >> 2, a_c = lane & 3, a_k = lane >> 4;
  float a[MF_QUADS];
  uint64_t badmask[MF_QUADS];
#pragma unroll
  for (int q = 0; q < MF_QUADS; q++) {
    const uint32_t h = hyp0 + 4 * q + a_hq;  // < ld_local: ld_local is a multiple of 256 >= MF_HYPS_PER_BLOCK blocks
    float v = 0.0f;
    if (a_c < 3) v = RtSoA[(size_t)(a_k < 3 ? 3 * a_c + a_k : 9 + a_c) * ld_local + h];
    a[q] = v;
    badmask[q] = __ballot(!(fabsf(v) < __builtin_inff()));  // lanes holding a non-finite coefficient
  }
  __syncthreads();
  uint32_t cnt[MF_QUADS];
#pragma unroll
  for (int q = 0; q < MF_QUADS; q++) cnt[q] = 0;
  const float* P4f = reinterpret_cast<const float*>(P4);
  const int col = lane & 15, kb = lane >> 4;
  for (int g = 0; g < padded; g += 16) {
    const float b = P4f[(g + col) * 4 + kb];
    const float4 cq = Qn[g + col];
    const f32x4 c = {cq.x, cq.y, cq.z, cq.w};
    // all MF_QUADS products first, into distinct result tiles (independent: they pipeline at the 32-cycle issue rate),
    // then the lane-local epilogues — a single reused tile would serialise on the 40-cycle result latency
    f32x4 d[MF_QUADS];
#pragma unroll
    for (int q = 0; q < MF_QUADS; q++) d[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b, c, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < MF_QUADS; q++) {
      const float d2 = fma_(d[q][2], d[q][2], fma_(d[q][1], d[q][1], d[q][0] * d[q][0]));
      cnt[q] += (d2 < tau2) ? 1u : 0u;
    }
  }
  // lane l counted hypothesis (quad q, member l >> 4) over the columns l & 15: sum the 16 columns
#pragma unroll
  for (int q = 0; q < MF_QUADS; q++) {
    uint32_t c = cnt[q];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 16);
    // coefficient lanes of member hq' = l >> 4 are {16 k + 4 hq' + c}: any non-finite one zeroes the hypothesis
    const uint64_t mine = 0x000F000F000F000Full << (4 * (lane >> 4));
    if (badmask[q] & mine) c = 0;
    const uint32_t h = hyp0 + 4 * q + (lane >> 4);
    if (col == 0) partial[(size_t)chunk * ld_local + h] = c;
  }
}

// One launch, two kinds of workgroup.  Hypotheses [0, hv) go to VALU workgroups (256 each), [hv, ld_local) to MFMA
// workgroups (128 each); the two kinds are interleaved along blockIdx.x so that every CU holds both and its vector
// and matrix pipes work at the same time (cdna guide: "MFMA and VALU pipes are separate").
// __launch_bounds__(256, 8): <= 64 VGPRs, 8 waves per SIMD — plain v_fma_f32 needs that occupancy on gfx950
// (measured 45 / 80 / 99 / 117 TFLOP/s at 1 / 2 / 4 / 8 waves per SIMD, tools/ubench_valu.hip).
template <int MODE>
__global__ __launch_bounds__(SCORE_THREADS, 8) void score_kernel(const float* __restrict__ planes, int n, int ld,
                                                                 const float* __restrict__ RtSoA, uint32_t ld_local,
                                                                 float tau2, int chunk_pts,
                                                                 uint32_t* __restrict__ partial, uint32_t nv,
                                                                 uint32_t nm) {
  __shared__ float4 smem[2 * SCORE_PC];  // 16 KiB: both bodies carve their point images out of it
  const uint32_t bx = blockIdx.x, tot = nv + nm;
  // Bresenham spread of the nm MFMA workgroups among the nv VALU ones
  const uint32_t m_before = (uint32_t)(((uint64_t)bx * nm) / tot), m_after = (uint32_t)(((uint64_t)(bx + 1) * nm) / tot);
  if (MODE == 0 && m_after != m_before)  // the matrix-pipe body counts inliers only
    score_mfma_body(smem, nv * SCORE_THREADS + m_before * MF_HYPS_PER_BLOCK, blockIdx.y, planes, n, ld, RtSoA, ld_local,
                    tau2, chunk_pts, partial);
  else
    score_valu_body<MODE>(smem, (bx - m_before) * SCORE_THREADS, blockIdx.y, planes, n, ld, RtSoA, ld_local, tau2, chunk_pts,
                    partial);
}

__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long k, unsigned long long* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(k, o);
    k = other > k ? other : k;
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = k;
  __syncthreads();
  unsigned long long b = lds[0];
  for (int w = 1; w < 4; w++) b = lds[w] > b ? lds[w] : b;
  return b;
}

// Per-hypothesis count (sum of the chunk partials) and the winner key pair, in ONE launch and without atomics on
// the keys: every block reduces its hypotheses to (best key, lowest position attaining it) and stores the pair; the
// block that takes the last ticket reduces the pairs and writes key2[0..1] (so key2 needs no zeroing).
//   key = (count << 32) | second, second = sel_key[g] or, without sel_key, 0xFFFFFFFF - g;  position = 0xFFFFFFFF - g.
// (r04 let the workgroup that takes the last ticket go on with the winner's own work — mask, rank index, (R, t) — instead of
// finalize_kernel's launch: bit-exact, and the arg-max launch went from 8.6 to 33 us at C2 against 6.4 us for that kernel's grid
// (one workgroup walks 5000 correspondences and 50 000 keys in ~70 dependent load rounds); removed in r05.)
__global__ __launch_bounds__(256) void score_argmax_kernel(const uint32_t* __restrict__ partial, uint32_t n_chunks,
                                                           Shard sh, const uint32_t* __restrict__ sel_key,
                                                           uint32_t* __restrict__ cnt_out,
                                                           unsigned long long* __restrict__ pairs,
                                                           uint32_t* __restrict__ ticket,
                                                           unsigned long long* __restrict__ key2) {
  __shared__ unsigned long long lds[4];
  __shared__ uint32_t s_last;
  // grid-stride over the hypotheses: at most 256 workgroups take a ticket (2000 same-address atomics cost ~20 us: C4)
  unsigned long long k = 0, pos = 0;
  for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < sh.n_local; l += gridDim.x * 256) {
    uint32_t c = 0;
    {  // (four rows in flight: a loop of n_chunks trips is n_chunks dependent round trips to hipcc — five at C2)
      uint32_t ch = 0;
      for (; ch + 4 <= n_chunks; ch += 4) {
        const uint32_t p0 = partial[(size_t)ch * sh.ld_local + l], p1 = partial[(size_t)(ch + 1) * sh.ld_local + l],
                       p2 = partial[(size_t)(ch + 2) * sh.ld_local + l], p3 = partial[(size_t)(ch + 3) * sh.ld_local + l];
        c += (p0 + p1) + (p2 + p3);
      }
      uint32_t pr[3];
#pragma unroll
      for (uint32_t u = 0; u < 3; u++) pr[u] = ch + u < n_chunks ? partial[(size_t)(ch + u) * sh.ld_local + l] : 0u;
      c += pr[0] + pr[1] + pr[2];
    }
    cnt_out[l] = c;
    const uint32_t g = shard_global_index(l, sh.block, sh.rank, sh.world);
    const uint32_t second = sel_key ? sel_key[g] : 0xFFFFFFFFu - g;
    if (c) {
      const unsigned long long kk = ((unsigned long long)c << 32) | (unsigned long long)second;
      const unsigned long long pp = (unsigned long long)(0xFFFFFFFFu - g);
      if (kk > k || (kk == k && pp > pos)) { k = kk; pos = pp; }
    }
  }
  const unsigned long long bk = block_max_u64(k, lds);
  __syncthreads();
  const unsigned long long bp = block_max_u64((k == bk) ? pos : 0ull, lds);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&pairs[2 * blockIdx.x], bk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pairs[2 * blockIdx.x + 1], bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // key2 == nullptr: the pairs go to this context's own finalize_kernel, whose workgroups each reduce them themselves (r05: the
  // release, the ticket and the last workgroup's pass over the pairs were ~2.5 us of this launch's 8.6 at C2)
  if (!key2) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1u : 0u;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  unsigned long long gk = 0, gp = 0;
  for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) {
    const unsigned long long a = __hip_atomic_load(&pairs[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long q = __hip_atomic_load(&pairs[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a > gk || (a == gk && q > gp)) { gk = a; gp = q; }
  }
  const unsigned long long K = block_max_u64(gk, lds);
  __syncthreads();
  const unsigned long long P = block_max_u64((gk == K) ? gp : 0ull, lds);
  if (threadIdx.x == 0) {
    key2[0] = K;
    key2[1] = (K != 0 && sel_key) ? P : 0ull;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

// ------------------------------------------------------------------------------------------------
// C2, inlier count, the default since r02: a matrix-pipe FILTER decides almost every test, an EXACT pass settles the rest.
//
// The count of hypothesis h is  #{m : d2(h, m) < tau^2}  with d2 the canonical fp32 chain (resid2, above).  Two kernels:
//
//   F  score_filter_kernel   evaluates  E' = 1024 s (R p + t - q)  for 8 hypotheses x 32 correspondences per
//      v_mfma_f32_32x32x16_f16: rows = (hypothesis, component) — three per hypothesis, the fourth idle — columns =
//      correspondences, K = 16 slots:
//          k0..8   (rh, rh, rl) x (Ph, Pl, Ph) for x, y, z     r = 1024 R and P = s p, each split in two fp16 halves
//          k9..14  -1024 x (Qh, Ql) of the row's own component  Q = s q
//          C       1024 s t (fp32)
//      (s: the power of two that puts the largest |coordinate| of the call into [256, 512).)  A lane then holds the three
//      components of ONE test; x = |E'|^2 - LO costs three fmas, its sign bit is shifted into a register (v_alignbit) and
//      counted once per 32 tests.  4.6 vector instructions per test instead of 17.
//      What F computes is NOT the canonical chain, so it only decides tests that are clear:
//          |E'| < 1024 (s tau - eta)  inlier        |E'| >= 1024 (s tau + eta)  outlier         else  UNDECIDED
//      eta bounds the distance between the two evaluations (below).  Undecided tests (about 1 in 10^4) go to a queue.
//   X  score_exact_kernel    evaluates the queued tests with the canonical chain and adds them to the counts (integer
//      atomics: order-free).  It also recounts, exactly, every (wave, split) F gave up on: a hypothesis outside the
//      filter's range (non-finite, |R_ij| > 1.5, huge t), tau too small or too large against the coordinate scale, a
//      queue that overflowed.  F never guesses; whatever it cannot bound it hands over.
//
// eta.  With |P|, |Q| <= Pmax, Qmax < 512, |t| s <= Tmax and sum_k |R_ck| <= 4.5, per component and in units of E'/1024:
//   fp16 splits   x = hi + lo + rem, |rem| <= 2^-22 |x| (or 2^-25 absolute below the fp16 normal range; inputs flushed
//                 to zero there would add < 1e-4): dropped rl Pl plus the two remainders <= 3.01 * 2^-22 sum|R| Pmax,
//                 and 2^-22 Qmax for q; the products themselves (11 x 11 bits) and C are exact in fp32;
//   accumulation  <= 17 additions, each off by <= 2^-23 of S = sum|R| Pmax + Qmax + Tmax (truncation allowed for);
//   the canonical chain itself: 4 roundings, <= 4 * 2^-24 S.
//   Sum <= 12.5 * 2^-22 S per component, sqrt(3) of it for the vector: < 2^-17.5 S.   eta = 2^-16 (2.6 Pmax + Qmax +
//   Tmax) >= 2^-16 S / 1.73 leaves a factor 1.7; LO and HI carry another 4e-6 for the roundings of the squares, of
//   sqrt(tau^2) and of d2 itself.  The parity suite compares every count with the canonical kernel's.
// ------------------------------------------------------------------------------------------------
// a launch that takes its dispatch packet's own start / stop timestamps only when asked to (SC_FLAG_TIMING_HOT): without events it is
// a plain launch
#define SC_LAUNCH_EV(kernel, grid, block, st, e0, e1, ...)                                              \
  do {                                                                                                  \
    if ((e0) != nullptr || (e1) != nullptr) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                   \
  } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float FX_RS = 1024.0f;
constexpr int FX_UNIT = 256;    // correspondences per staging unit (8 MFMA steps); LDS holds two
constexpr int FX_WIN = 1024;    // correspondences per window: 32 steps, one 32-bit shift register per test
constexpr int FX_NQ = 256;      // global sub-queues (one ticket counter for ~6000 waves costs ~50 us of same-address atomics)
constexpr int FX_QL = 256;      // LDS queue entries per wave
constexpr int FX_WAVES = 4;     // waves per workgroup, 8 hypotheses each
struct FilterInfo { float s, pmax, qmax, pad; };
constexpr size_t FX_INFO_BYTES = 512;

struct FilterState {  // device view of the state buffer (filter_plan().state_bytes)
  FilterInfo* info;   // written by the tile kernel
  uint32_t* qcount;   // FX_NQ ticket counters, one per 128-byte line
  uint32_t* redo;     // bit (split * n_waves + wave): X recounts the wave's 8 hypotheses over the split exactly
  uint2* queue;       // FX_NQ sub-queues of cap_sq entries {correspondence, wave << 5 | lane half << 4 | tests}
  uint32_t cap_sq;
  uint32_t zero_words;  // counters + bitmap, cleared by the tile kernel
};
static FilterState filter_state(void* state, const FilterPlan& fp) {
  FilterState f;
  unsigned char* p = static_cast<unsigned char*>(state);
  f.info = reinterpret_cast<FilterInfo*>(p); p += FX_INFO_BYTES;
  f.qcount = reinterpret_cast<uint32_t*>(p); p += (size_t)FX_NQ * 128;
  f.redo = reinterpret_cast<uint32_t*>(p);
  const size_t bm_words = ((size_t)fp.splits * fp.n_waves + 31) / 32;
  p += (bm_words * 4 + 127) / 128 * 128;
  f.queue = reinterpret_cast<uint2*>(p);
  f.cap_sq = fp.queue_cap / FX_NQ;
  f.zero_words = (uint32_t)(FX_NQ * 32 + bm_words);
  return f;
}

// Which kernel.  The filters pay a tile kernel, an exact pass and a few microseconds of set-up per workgroup: below ~1.3e8
// tests the plain kernel is as fast or faster (C1, 2e7 tests: 9 us plain, 20 us filtered).  Big calls take the Gram filter
// wherever even a hypothesis FAR from the call's reference frame keeps its shell inside tau'^2 (all four BASELINE scenes since
// r04b; until then tau had to be > ~2 % of the clouds' extent, which left C3 to the linear filter), else the linear one.
int score_filter_mode(int score_mode, const Tuning& tn, int n, uint32_t ld_local, uint64_t host_max, const uint64_t* host_box,
                      float tau2) {
  if (score_mode != 0 || tn.score_filter == 1 || tn.score_split != 0) return 0;
  if (ld_local == 0 || ld_local / 8 >= (1u << 27)) return 0;
  // beyond 2^36 tests the queue of undecided tests (sized T n / 512 entries: ~20x what the BASELINE scenes need) would
  // pass 1 GB: such calls keep the plain kernel rather than a queue that may overflow into wholesale recounts
  const uint64_t tests = (uint64_t)ld_local * (uint64_t)n;
  if (tests > (1ull << 36)) return 0;
  const bool big = tests >= (1ull << 27);
  if (tn.score_filter == 2) return 1;
  const bool gram_fits = ld_local <= (1u << 20) && n <= (1 << 24);  // (the queue entry carries wave << 17)
  if (tn.score_filter == 3) return gram_fits ? 2 : 1;
  if (!big) return 0;
  if (host_max == ~0ull) return 1;  // statistics not known: the linear filter sorts itself out (it recounts what it cannot bound)
  if (gram_fits && host_box) {
    // Gram: even a hypothesis FAR from the call's reference frame (dM of size 1, tau' a quarter of the clouds, every
    // correspondence looked at) must keep its shell well inside tau'^2 — the NEAR ones are two orders of magnitude better off.
    // The frame itself is voted on the device later; here it is taken to map one box's centre onto the other's.
    double hP2 = 0.0, hQ2 = 0.0, hmax = 0.0;
    for (int c = 0; c < 6; c++) {
      const float mx = float_unkey((uint32_t)host_box[c]), mn = -float_unkey((uint32_t)(host_box[c] >> 32));
      const float ctr = 0.5f * mx + 0.5f * mn;
      const double a = (double)mx - (double)ctr, b = (double)ctr - (double)mn, h = a > b ? a : b;
      (c < 3 ? hP2 : hQ2) += h * h;
      if (c < 3 && h > hmax) hmax = h;
    }
    const double hQn = sqrt(hQ2);
    hmax = hQn > hmax ? hQn : hmax;
    union { uint32_t u; float f; } pa, qa;
    pa.u = (uint32_t)host_max; qa.u = (uint32_t)(host_max >> 32);
    if (hmax > 0.0 && hmax < 1e30) {
      int e = 0;
      (void)frexp(hmax, &e);  // hmax in [2^(e-1), 2^e)
      const double s = ldexp(1.0, 6 - e), Pn = s * sqrt(hP2), Qn = s * hQn, st = s * sqrt((double)tau2);
      const double eps = gram_eps(2.0, 1.0, 0.25 * (Pn + Qn), 3e-7, Pn, Qn, Pn + Qn, st);
      const double dE = GX_CANON * s * ((double)qa.f + 1.75 * (double)pa.f + (double)qa.f);
      if (eps <= 0.2 * st * st && dE <= 0.02 * st && st <= 32.0 && st >= 0.2) return 2;
    }
  }
  return filter_in_range(host_max, tau2) ? 1 : 0;
}

bool filter_in_range(uint64_t host_max, float tau2) {
  if (host_max == ~0ull) return true;
  union { uint32_t u; float f; } a, b, m;
  a.u = (uint32_t)host_max; b.u = (uint32_t)(host_max >> 32);
  const float pmax = a.f, qmax = b.f;
  m.f = pmax > qmax ? pmax : qmax;
  const uint32_t mb = m.u;
  int k = 8 - ((int)((mb >> 23) & 255u) - 127);
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float s = ldexpf(1.0f, k), st = s * sqrtf(tau2);
  const float eta = (2.6f * (pmax * s) + qmax * s) * (1.0f / 65536.0f);  // a wave adds its own |t| s to this
  return (eta <= 0.25f * st) && (st <= 4096.f) && (pmax * s < 512.f) && (qmax * s < 512.f);  // false on NaN
}

FilterPlan filter_plan(int n, uint32_t ld_local, const Tuning& tn, uint32_t mode) {
  FilterPlan fp;
  fp.mode = mode;
  fp.windows = (uint32_t)((n + FX_WIN - 1) / FX_WIN);
  fp.n_waves = ld_local / 8;  // recount units: 8 hypotheses (a wave of the linear filter, a quarter of a Gram wave)
  // grid.y: the windows are split so that the launch has several generations of workgroups and the last one is well filled
  // (linear: 32 hypotheses per workgroup, 6 resident per CU; Gram: 256 hypotheses, 2 resident)
  const uint32_t groups = mode == 2 ? (ld_local + 32 * GX_WAVES - 1) / (32 * GX_WAVES) : ld_local / (8 * FX_WAVES);
  const uint32_t slots = mode == 2 ? 256 * 2 : 256 * 6;
  uint32_t best = 1; double best_eff = 0.0;
  const uint32_t smax = fp.windows < 8 ? fp.windows : 8;
  for (uint32_t sp = 1; sp <= smax; sp++) {
    const uint32_t per = (fp.windows + sp - 1) / sp, used = (fp.windows + per - 1) / per;  // splits that get windows
    if (used != sp) continue;
    const double blocks = (double)groups * sp, gens = (double)((uint64_t)(blocks + slots - 1) / slots);
    // every workgroup pays its set-up once: ~6 % per extra split (C2: 71 / 61 / 61 / 66 us at 1 / 2 / 3 / 5 splits,
    // C4: 476 / 492 / 514 / 555)
    const double eff = blocks / (gens * slots) * (1.0 - 0.06 * (sp - 1));
    if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
  }
  fp.splits = tn.filter_splits && tn.filter_splits <= fp.windows ? tn.filter_splits : best;
  {  // a forced value must still give every split at least one window
    const uint32_t per = (fp.windows + fp.splits - 1) / fp.splits;
    fp.splits = (fp.windows + per - 1) / per;
  }
  if (mode == 2) {
    // The Gram filter cuts in UNITS (gram_geom), and only its rows FAR from the call's frame — a tenth of them on the BASELINE
    // scenes — walk every unit.  Their workgroups must not outlast everything else (C3 at 3 splits of 27 units: 113 of the
    // kernel's 120 us were those workgroups' own length), but every split is another workgroup per group of rows with its own
    // set-up (C4: 112 / 127 / 138 / 150 us at 1 / 3 / 5 / 8 splits).  So: the fewest splits that keep a far workgroup's walk —
    // ~4.2 us per unit — under half of what the whole launch should take if 15 % of the rows are far and a fifth of the
    // correspondences near (0.32 units' worth per group and unit, over 512 resident workgroups), and never below 4 units.
    // C2 5 splits (4 units), C3 4 (20), C4 1 (20).  (Empty workgroups cost nothing: tools/ubench/dispatch_rate.hip, 0.25 ns apiece.)
    const uint32_t units = fp.windows * (uint32_t)(FX_WIN / GX_UNIT);
    // r04c: + 2.5 units' worth of set-up per workgroup in the launch's length, and 0.6 of it instead of half — C4 (1954 row blocks of
    // 20 units) is a launch of ~34 units, most of it the near workgroups' fixed cost; its far workgroups' 20 units fit in it whole:
    // 1 split 98 us, 2 splits 108, 3 108, 4 112, 8 131 (C3: 1 167, 3 149, 4 133, 5 131, 8 130 — it takes 4 now; C2 unchanged: 5)
    const double launch_units = 0.6 * (double)groups * (2.5 + 0.32 * (double)units) / 512.0;  // 0.6 of the launch, in units of one workgroup's walk
    const uint32_t walk = launch_units < 4.0 ? 4u : (uint32_t)launch_units;
    uint32_t sp = tn.filter_splits ? tn.filter_splits : (units + walk - 1) / walk;
    sp = sp < 1u ? 1u : (sp > 8u ? 8u : sp);
    if (sp > units) sp = units;
    const uint32_t pu = (units + sp - 1) / sp;
    fp.splits = (units + pu - 1) / pu;  // every split gets at least one unit
  }
  fp.rows = fp.windows * FX_WIN + (mode == 2 ? GX_UNIT : FX_UNIT);
  uint64_t cap = (uint64_t)ld_local * (uint64_t)n / 512;  // ~20x what the BASELINE scenes queue
  if (cap < (1u << 16)) cap = 1u << 16;
  if (cap > (1u << 27)) cap = 1u << 27;
  if (tn.filter_queue_cap) cap = tn.filter_queue_cap;
  fp.queue_cap = (uint32_t)(cap / FX_NQ * FX_NQ);
  if (fp.queue_cap < FX_NQ) fp.queue_cap = FX_NQ;
  fp.tile_bytes = (size_t)fp.rows * (mode == 2 ? GX_TILE_B : 32);
  const size_t bm_words = ((size_t)fp.splits * fp.n_waves + 31) / 32;
  fp.state_bytes = FX_INFO_BYTES + (size_t)FX_NQ * 128 + (bm_words * 4 + 127) / 128 * 128 + (size_t)fp.queue_cap * 8;
  // Gram: per coefficient row 64 bytes of fp16 halves + C, W, flag, hperm; per tile row pperm
  fp.coef_bytes = mode == 2 ? (size_t)ld_local * (64 + 16) + (size_t)fp.rows * 4 : 0;
  return fp;
}

// The fp16 image of the correspondences, 32 bytes each, in the K order of the B operand:
//   [Pxh Pxl Pxh  Pyh Pyl Pyh  Pzh Pzl | Pzh  Qxh Qxl  Qyh Qyl  Qzh Qzl  0];  rows [n, rows) are sentinels (far away).
// The two 16-byte halves of a correspondence are NOT adjacent: see the group layout at the end of filter_tile_block.
// mx_cur: max |p| and max |q| of the call (bit patterns; stage_points_kernel's atomicMax).  Also clears the filter's
// counters and bitmap for this call, so nothing needs a memset.
__device__ void gram_tile_block(const float* __restrict__ planes, int n, int ld, const FilterTileJob& job, uint32_t block,
                                uint32_t blocks);
__device__ void filter_tile_block(const float* __restrict__ planes, int n, int ld, const FilterTileJob& job, uint32_t block,
                                  uint32_t blocks) {
  if (job.mode == 2) { gram_tile_block(planes, n, ld, job, block, blocks); return; }
  const uint32_t m = block * 256 + threadIdx.x;
  for (uint32_t z = m; z < job.zero_words; z += blocks * 256) job.zero[z] = 0u;
  const float Pmax = __uint_as_float(job.mx_cur[0]), Qmax = __uint_as_float(job.mx_cur[1]), mxv = fmaxf(Pmax, Qmax);
  const int e = (int)((__float_as_uint(mxv) >> 23) & 255u) - 127;  // mxv in [2^e, 2^(e+1))
  int k = 8 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  const float s = __uint_as_float((uint32_t)(k + 127) << 23);
  if (m == 0) *static_cast<FilterInfo*>(job.info) = FilterInfo{s, Pmax, Qmax, 0.f};
  if (m >= job.rows) return;
  _Float16 hi[6], lo[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const float X = m < (uint32_t)n ? planes[(size_t)c * ld + m] * s : (c < 3 ? 0.f : 32768.f);
    hi[c] = (_Float16)X;
    lo[c] = (_Float16)(X - (float)hi[c]);
  }
  half8 f0 = {hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2]};
  half8 f1 = {hi[2], hi[3], lo[3], hi[4], lo[4], hi[5], lo[5], (_Float16)0.f};
  // per group of 32 correspondences (one MFMA step): the 32 first halves (k0..7), then the 32 second halves (k8..15) —
  // lane l of a wave then reads the 16 bytes at 16 l of the group's 1 KiB: a linear, bank-conflict-free ds_read_b128
  // (with the two halves of a correspondence side by side the lanes of a half stride 32 bytes: 2-way conflicts, +4
  // cycles on each 8-cycle read by SQ_LDS_BANK_CONFLICT)
  uint4* tile = static_cast<uint4*>(job.tile);
  const size_t slot = (size_t)(m >> 5) * 64 + (m & 31u);
  tile[slot] = *reinterpret_cast<uint4*>(&f0);
  tile[slot + 32] = *reinterpret_cast<uint4*>(&f1);
}
__global__ __launch_bounds__(256) void filter_tile_kernel(const float* __restrict__ planes, int n, int ld, FilterTileJob job) {
  filter_tile_block(planes, n, ld, job, blockIdx.x, gridDim.x);
}

// F.  Workgroup = FX_WAVES waves; wave w of workgroup b owns hypotheses 8 (FX_WAVES b + w) .. + 7 and never talks to
// the others except through the B tile: a unit of 256 correspondences (8 KiB) is brought into LDS by LDS-DMA while
// the previous one is being used (two buffers, one barrier per unit).  blockIdx.y = split of the windows.
// 80 VGPRs: six waves per SIMD cover the MFMA and LDS latencies, so nothing is double-buffered inside a wave.
// VAR (Tuning::filter_variant): 0 = one MFMA, then its epilogue (the wave idles through the MFMA's latency; the other
// waves of the SIMD fill it); bit 0 = the NEXT step's MFMA is issued before this step's epilogue (two accumulator sets,
// 96 VGPRs, 5 waves per SIMD): vector and matrix work of ONE wave overlap; bit 1 = s_setprio 1 around the MFMA issue;
// 16 / 32: timing-only ablations (no MFMA / no epilogue: wrong counts) for tools/ab_stage.py.
template <int WAVES, int VAR>
__global__ __launch_bounds__(64 * WAVES, (VAR & 128) ? 8 : ((VAR & 1) ? 5 : 6)) void score_filter_kernel(const float* __restrict__ Rt, uint32_t ldl, float tau2,
                                                                     const uint4* __restrict__ tile,
                                                                     const FilterInfo* __restrict__ info, uint32_t windows,
                                                                     uint32_t splits, uint32_t n_waves,
                                                                     uint32_t* __restrict__ cnt_out, uint2* __restrict__ gq,
                                                                     uint32_t cap_sq, uint32_t* __restrict__ qcount,
                                                                     uint32_t* __restrict__ redo_bits, uint32_t ql) {
  __shared__ uint4 Bt[2][FX_UNIT * 2];
  __shared__ float4 Ttab[WAVES][8];
  constexpr int QL = (VAR & 128) ? FX_QL / 2 : FX_QL;  // (128: 8 workgroups per CU — their LDS must fit 160 KiB)
  __shared__ uint32_t queue[WAVES][QL];
  if (ql > (uint32_t)QL) ql = QL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hf = lane >> 5;
  const uint32_t per = (windows + splits - 1) / splits, w0 = blockIdx.y * per, w1 = min(windows, w0 + per);
  const uint32_t wid = blockIdx.x * WAVES + wave, h0 = wid * 8;
  const uint32_t u0 = w0 * (FX_WIN / FX_UNIT), u1 = w1 * (FX_WIN / FX_UNIT);  // staging units of this workgroup
  // The DMA is issued through asm so that hipcc does not count it: with the builtin the compiler waits vmcnt(0) before
  // the NEXT ds_read (it cannot tell the two buffers apart), which exposes the whole DMA latency in every unit.  Our own
  // wait sits before the barrier that publishes the buffer.  (Uncounted loads only make the compiler's own vmcnt(N)
  // waits stronger: the counter retires in order.)
  auto stage = [&](uint32_t u, int buf) {
#pragma unroll
    for (int i = 0; i < 8 / WAVES; i++) {
      const uint4* gsrc = tile + (size_t)u * (FX_UNIT * 2) + 64 * WAVES * i + tid;
      const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&Bt[buf][64 * WAVES * i + wave * 64]));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  if (u0 < u1) stage(u0, 0);
  const FilterInfo fi = *info;
  // A fragment: row r = (hypothesis r >> 2, component r & 3); lane half 0 holds k0..7, half 1 k8..15
  half8 A;
  {
    const int r = lane & 31, hy = r >> 2, c = r & 3;
    float x[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) x[kk] = Rt[(size_t)(3 * (c < 3 ? c : 0) + kk) * ldl + h0 + hy];  // unconditional: one round trip
    _Float16 rh[3], rl[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const float v = c < 3 ? x[kk] * FX_RS : 0.f;
      rh[kk] = (_Float16)v;
      rl[kk] = (_Float16)(v - (float)rh[kk]);
    }
    const _Float16 z = (_Float16)0.f, mone = (_Float16)(-FX_RS);
    const half8 a0 = {rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2]};
    const half8 a1 = {rl[2], c == 0 ? mone : z, c == 0 ? mone : z, c == 1 ? mone : z, c == 1 ? mone : z, c == 2 ? mone : z, c == 2 ? mone : z, z};
    A = hf ? a1 : a0;
  }
  float tmax = 0.f;
  bool wild = false;
  if (lane < 8) {
    const uint32_t h = h0 + lane;
    float v[12];
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = Rt[(size_t)c * ldl + h];  // all twelve in flight together
    float rmax = 0.f, nanp = 0.f;
#pragma unroll
    for (int c = 0; c < 9; c++) rmax = fmaxf(rmax, fabsf(v[c]));
    tmax = fmaxf(fabsf(v[9]), fmaxf(fabsf(v[10]), fabsf(v[11])));
#pragma unroll
    for (int c = 0; c < 12; c++) nanp += v[c] * 0.f;  // NaN or infinity anywhere poisons the sum (fmaxf drops NaNs)
    wild = !(rmax <= 1.5f) || !(tmax < 1e30f) || !(nanp == 0.f);
    Ttab[wave][lane] = make_float4(v[9] * FX_RS * fi.s, v[10] * FX_RS * fi.s, v[11] * FX_RS * fi.s, 0.f);
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  tmax = __shfl(tmax, 0);
  const bool any_wild = __ballot(wild) != 0;
  const float s = fi.s, st = s * sqrt_rn(tau2);
  const float Sb = 2.6f * (fi.pmax * s) + fi.qmax * s + tmax * s;
  const float eta = Sb * (1.0f / 65536.0f);
  // the filter's range (every comparison is false on a NaN): the shell must stay thin against tau, the scaled
  // quantities inside what the error bound assumes
  const bool fast = !any_wild && (eta <= 0.25f * st) && (st <= 4096.f) && (tmax * s <= 2048.f) && (fi.pmax * s < 512.f) &&
                    (fi.qmax * s < 512.f);
  const float lo_e = FX_RS * (st - eta), hi_e = FX_RS * (st + eta);
  const float LO = lo_e * lo_e * (1.0f - 4e-6f), HI = hi_e * hi_e * (1.0f + 4e-6f);
  const uint32_t W2b = fast ? __float_as_uint(HI - LO) : 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // Ttab: written and read by this wave only
  f32x16 C;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const float4 T = Ttab[wave][2 * jj + hf];
    C[4 * jj] = T.x; C[4 * jj + 1] = T.y; C[4 * jj + 2] = T.z; C[4 * jj + 3] = 0.f;
  }
  uint32_t total[4] = {0, 0, 0, 0};
  uint32_t sr[4] = {0, 0, 0, 0};
  uint32_t qn = 0;
  bool redo = !fast;
  uint32_t* q = queue[wave];
  auto flush = [&]() {  // the wave's queue -> its global sub-queue (one ticket)
    const uint32_t sq = wid % FX_NQ;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&qcount[sq * 32], qn);
    base = __shfl(base, 0);
    // a ticket that does not fit leaves no hole: the part of it inside the sub-queue is filled with empty entries (the
    // exact pass reads min(count, capacity) of them), and the wave asks for a recount
    const bool fits = base + qn <= cap_sq;
    for (uint32_t i = lane; i < qn && base + i < cap_sq; i += 64)
      gq[(size_t)sq * cap_sq + base + i] = fits ? make_uint2(q[i] >> 5, (wid << 5) | (q[i] & 31u)) : make_uint2(0u, 0u);
    if (!fits) redo = true;
    qn = 0;
  };
  for (uint32_t u = u0; u < u1; u++) {
    const int buf = (int)((u - u0) & 1u);
    if constexpr ((VAR & 256) == 0) {                    // (256: timing-only ablation, no barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of unit u has landed
      __syncthreads();                                   // ... everybody's has, and the other buffer is free
    }
    if (u + 1 < u1) stage(u + 1, buf ^ 1);
    if (!redo) {
      const half8* Bc = reinterpret_cast<const half8*>(Bt[buf]) + lane;  // (tile layout: filter_tile_block)
      auto mfma = [&](const half8& b) -> f32x16 {
        if constexpr ((VAR & 16) != 0) {  // ablation: no matrix instruction (the operand stays live)
          f32x16 D = C;
          asm volatile("" ::"v"(b));
#pragma unroll
          for (int i = 0; i < 16; i++) asm volatile("" : "+v"(D[i]));  // opaque: nothing of the epilogue can be hoisted
          return D;
        } else {
          if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);
          const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, b, C, 0, 0, 0);
          if constexpr ((VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);
          return D;
        }
      };
      // the vector work of one step: x = |E'|^2 - LO per test, sign bit into the shift register, shell test, queue
      auto epilogue = [&](const f32x16& D, int g) {
        if constexpr ((VAR & 32) != 0) {  // ablation: no epilogue
          sr[0] ^= __float_as_uint(D[0]) ^ __float_as_uint(D[5]) ^ __float_as_uint(D[10]) ^ __float_as_uint(D[14]);
          return;
        }
        float x[4];
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const float v = fma_(D[4 * jj + 2], D[4 * jj + 2], fma_(D[4 * jj + 1], D[4 * jj + 1], fma_(D[4 * jj], D[4 * jj], -LO)));
          x[jj] = v;
          sr[jj] = __builtin_amdgcn_alignbit(sr[jj], __float_as_uint(v), 31);  // sign bit: |E'|^2 < LO, a certain inlier
          mn = min(mn, __float_as_uint(v));
        }
        if constexpr ((VAR & 512) != 0) { sr[0] ^= mn; return; }  // (512: timing-only ablation, no shell test)
        const uint64_t hm = __ballot(mn < W2b);  // as unsigned integers: 0 <= x < HI - LO, the undecided shell
        if (__builtin_expect(hm != 0, 0)) {
          const uint32_t k2 = (uint32_t)__popcll(hm);
          if (qn + k2 <= ql) {
            uint32_t bits = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) bits |= (__float_as_uint(x[t]) < W2b) ? (1u << t) : 0u;
            if (mn < W2b) q[qn + __popcll(hm & ((1ull << lane) - 1ull))] = ((u * FX_UNIT + 32 * g + col) << 5) | ((uint32_t)hf << 4) | bits;
            qn += k2;
          } else {
            redo = true;  // more undecided tests than the queue holds: the exact pass takes the whole (wave, split)
          }
        }
        // The idle fourth rows of D stay "in use" up to here: otherwise the register allocator parks this epilogue's
        // temporaries in the idle rows of the OTHER accumulator set — whose MFMA is still in flight — and has to pad the
        // write-after-write hazard with s_nop until that MFMA has finished, which is the overlap this variant is after.
        if constexpr ((VAR & 1) != 0) asm volatile("" ::"v"(D[3]), "v"(D[7]), "v"(D[11]), "v"(D[15]));
      };
      if constexpr ((VAR & 1) != 0) {
        // software pipeline inside the unit: step g + 1's MFMA is in the matrix pipe while step g's epilogue issues
        constexpr int G = FX_UNIT / 32;
        half8 b0 = Bc[0], b1 = Bc[64];
        f32x16 D0 = mfma(b0), D1;
#pragma unroll 1
        for (int g = 0; g < G; g += 2) {
          D1 = mfma(b1);                                   // step g + 1
          if (g + 2 < G) b0 = Bc[64 * (g + 2)];
          epilogue(D0, g);
          if (g + 2 < G) {
            D0 = mfma(b0);                                 // step g + 2
            b1 = Bc[64 * (g + 3)];
          }
          epilogue(D1, g + 1);
        }
      } else {
        half8 b = Bc[0];
#pragma unroll 2  // (by 4: no gain; fully unrolled: 144 bytes of spills inside the loop, 64 -> 118 us on C2)
        for (int g = 0; g < FX_UNIT / 32; g++) {
          const f32x16 D = mfma(b);
          if constexpr ((VAR & 64) == 0) {  // (64: ablation, the operand is not re-read)
            if (g + 1 < FX_UNIT / 32) b = Bc[64 * (g + 1)];
          }
          epilogue(D, g);
        }
      }
      if ((u + 1) % (FX_WIN / FX_UNIT) == 0) {  // window boundary: 32 tests per register
#pragma unroll
        for (int t = 0; t < 4; t++) { total[t] += (uint32_t)__popc(sr[t]); sr[t] = 0; }
      }
      if (qn > ql / 2) flush();
    }
  }
  if (qn && !redo) flush();
  if (redo && lane == 0) {
    const size_t bit = (size_t)blockIdx.y * n_waves + wid;
    atomicOr(&redo_bits[bit >> 5], 1u << (bit & 31));
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {
    uint32_t c = total[t];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) c += __shfl_xor(c, o, 32);
    if (col == 0) cnt_out[(size_t)blockIdx.y * ldl + h0 + 2 * t + hf] = redo ? 0u : c;
  }
}

// How the Gram filter's launch cuts the tile's units (256 correspondences) into the `splits` workgroups of a group of 256 rows —
// shared by the filter and the exact pass (which must find the split a queued test belongs to).
//   rows not all near the frame: split y walks units [y pu_far, (y + 1) pu_far) of ALL the units;
//   rows all near the frame:     only `ns` splits work — split y walks [y pu_near, (y + 1) pu_near) of the NEAR units —, and ns grows
//                                beyond 1 only while the launch has fewer workgroups than the chip has slots (C2: 196 groups).
struct GramGeom { uint32_t units, near_units, pu_far, ns, pu_near; };
__device__ __forceinline__ GramGeom gram_geom(uint32_t windows, uint32_t splits, uint32_t groups, const GramFrame* __restrict__ fr) {
  GramGeom g;
  g.units = windows * (uint32_t)(FX_WIN / GX_UNIT);
  g.near_units = min(g.units, ((uint32_t)fr->pts + GX_UNIT - 1) / GX_UNIT);
  g.pu_far = (g.units + splits - 1) / splits;
  const uint32_t room = 512u / max(groups, 1u);
  g.ns = max(1u, min(min(splits, max(g.near_units, 1u)), room));
  g.pu_near = (max(g.near_units, 1u) + g.ns - 1) / g.ns;
  return g;
}
__device__ __forceinline__ bool gram_near_block(uint32_t bx, uint32_t S, const GramFrame* __restrict__ fr) {
  return (bx / S + 1u) * (32u * GX_WAVES) <= (uint32_t)fr->seg[bx % S].cnt;  // all 256 rows are hypotheses near the frame
}

// X.  First the queued tests (one per thread; workgroup b serves sub-queue b % FX_NQ), then the (wave, split) pairs F
// gave up on (one per workgroup at a time).  Everything here is the canonical chain; the counts are integers.
__device__ __forceinline__ void load_rt_aos(const float4* __restrict__ RtAoS, uint32_t h, float (&M)[12]) {
  const float4 a = RtAoS[3 * (size_t)h], b = RtAoS[3 * (size_t)h + 1], c = RtAoS[3 * (size_t)h + 2];
  M[0] = a.x; M[1] = a.y; M[2] = a.z; M[3] = a.w; M[4] = b.x; M[5] = b.y; M[6] = b.z; M[7] = b.w;
  M[8] = c.x; M[9] = c.y; M[10] = c.z; M[11] = c.w;
}
__global__ __launch_bounds__(256) void score_exact_kernel(const float* __restrict__ planes, int n, int ld,
                                                          const float4* __restrict__ RtAoS, uint32_t ldl, float tau2,
                                                          uint32_t windows, uint32_t splits, uint32_t n_waves,
                                                          const uint2* __restrict__ gq, uint32_t cap_sq,
                                                          const uint32_t* __restrict__ qcount,
                                                          const uint32_t* __restrict__ redo_bits,
                                                          uint32_t* __restrict__ cnt_out,
                                                          const uint32_t* __restrict__ hperm,
                                                          const uint32_t* __restrict__ pperm,
                                                          const GramFrame* __restrict__ fr) {
  // hperm / pperm (the Gram filter; null for the linear one): its rows are PERMUTED — coefficient row -> hypothesis, tile row ->
  // correspondence.  Queue entries and recount bits speak of rows; the counts and the canonical chain of hypotheses and correspondences.
  const bool gram = hperm != nullptr;
  const uint32_t per = (windows + splits - 1) / splits;
  // the Gram filter's splits are cut in units, differently for rows near the frame and the others (gram_geom)
  const uint32_t groups = ldl / (32u * GX_WAVES), S = gram_segments(groups);
  GramGeom gg{};
  if (gram) gg = gram_geom(windows, splits, groups, fr);
  const float4* __restrict__ aos4 = reinterpret_cast<const float4*>(planes + 6 * (size_t)ld);  // 8 floats per correspondence
  const uint32_t sq = blockIdx.x % FX_NQ, nq = min(qcount[sq * 32], cap_sq);
  for (uint32_t i = (blockIdx.x / FX_NQ) * 256 + threadIdx.x; i < nq; i += (gridDim.x / FX_NQ) * 256) {
    const uint2 e = gq[(size_t)sq * cap_sq + i];
    if (gram) {  // {tile row, wave of 32 rows << 17 | lane half << 16 | one bit per accumulator register}
      const uint32_t mr = e.x, w32 = e.y >> 17, ehf = (e.y >> 16) & 1u;
      const uint32_t sp = (mr / (uint32_t)GX_UNIT) / (gram_near_block(w32 / GX_WAVES, S, fr) ? gg.pu_near : gg.pu_far);
      const uint32_t m = pperm[mr];
      const float4 pa = aos4[2 * (size_t)m], pb = aos4[2 * (size_t)m + 1];
      for (uint32_t bits = e.y & 0xFFFFu; bits; bits &= bits - 1) {
        const uint32_t i16 = (uint32_t)(__ffs(bits) - 1), row = 8 * (i16 >> 2) + 4 * ehf + (i16 & 3u), hr = w32 * 32 + row;
        const size_t bit = (size_t)sp * n_waves + (hr >> 3);
        if ((redo_bits[bit >> 5] >> (bit & 31)) & 1u) continue;  // recounted as a whole below
        const uint32_t h = hperm[hr];
        float M[12];
        load_rt_aos(RtAoS, h, M);
        const float d2 = resid2(M, pa.x, pa.y, pa.z, pa.w, pb.x, pb.y);
        if (finite12(M) && d2 < tau2) atomicAdd(&cnt_out[(size_t)sp * ldl + h], 1u);
      }
      continue;
    }
    const uint32_t m = e.x, wid = e.y >> 5, ehf = (e.y >> 4) & 1u, sp = (m / FX_WIN) / per;
    const size_t bit = (size_t)sp * n_waves + wid;
    if ((redo_bits[bit >> 5] >> (bit & 31)) & 1u) continue;  // recounted as a whole below
    const float4 pa = aos4[2 * (size_t)m], pb = aos4[2 * (size_t)m + 1];  // the correspondence in two 16-byte loads
    for (uint32_t bits = e.y & 0xFu; bits; bits &= bits - 1) {
      const uint32_t h = wid * 8 + 2 * (uint32_t)(__ffs(bits) - 1) + ehf;
      float M[12];
      load_rt_aos(RtAoS, h, M);
      const float d2 = resid2(M, pa.x, pa.y, pa.z, pa.w, pb.x, pb.y);
      if (finite12(M) && d2 < tau2) atomicAdd(&cnt_out[(size_t)sp * ldl + h], 1u);
    }
  }
  const size_t nbits = (size_t)splits * n_waves;
  for (size_t w = blockIdx.x; w < (nbits + 31) / 32; w += gridDim.x) {
    uint32_t word = redo_bits[w];
    while (word) {
      const size_t bit = w * 32 + (size_t)(__ffs(word) - 1);
      word &= word - 1;
      const uint32_t sp = (uint32_t)(bit / n_waves), wid = (uint32_t)(bit % n_waves);
      const uint32_t hr = wid * 8 + (threadIdx.x & 7), h = gram ? hperm[hr] : hr;
      float M[12];
      load_rt_aos(RtAoS, h, M);
      const bool ok = finite12(M);
      uint32_t m0 = sp * per * FX_WIN, m1 = min((uint32_t)n, min(windows, (sp + 1) * per) * FX_WIN);
      if (gram) {  // the units its filter workgroup walked (rows near the frame: of the near units only — the rest cannot hold an inlier)
        const bool nb = gram_near_block(wid / 32u, S, fr);
        const uint32_t pu = nb ? gg.pu_near : gg.pu_far, ue = nb ? gg.near_units : gg.units;
        m0 = sp * pu * (uint32_t)GX_UNIT;
        m1 = min((uint32_t)n, min(ue, (sp + 1) * pu) * (uint32_t)GX_UNIT);
      }
      uint32_t cnt = 0;
      for (uint32_t mr = m0 + (threadIdx.x >> 3); mr < m1; mr += 32) {
        const uint32_t m = gram ? pperm[mr] : mr;
        const float4 pa = aos4[2 * (size_t)m], pb = aos4[2 * (size_t)m + 1];
        const float d2 = resid2(M, pa.x, pa.y, pa.z, pa.w, pb.x, pb.y);
        cnt += (ok && d2 < tau2) ? 1u : 0u;
      }
      cnt += __shfl_xor(cnt, 8);
      cnt += __shfl_xor(cnt, 16);
      cnt += __shfl_xor(cnt, 32);
      if ((threadIdx.x & 63) < 8 && cnt) atomicAdd(&cnt_out[(size_t)sp * ldl + h], cnt);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// C2, inlier count, the GRAM filter (r03): the matrix pipe evaluates the squared residual itself — since r04b in the FRAME OF A
// REFERENCE HYPOTHESIS, which (a) makes the cancelling terms small exactly where precision is needed and (b) lets most tests be
// skipped with a proof instead of being evaluated.
//
// The frame.  gram_ref_kernel lets GX_VOTE hypotheses spread over the ranked list vote for the one most of them agree with
// (R0, t0; R0 rebuilt orthogonal in fp64 — ANY rotation is a valid frame, a good one is only faster).  With c the centre of the
// source cloud's box and s a power of two,
//     P' = s (p - c),   Q' = s (R0^T (q - t0) - c),   V' = Q' - P'          (per correspondence; every |coordinate| < 64)
//     M  = R0^T R_h,    dM = M - I,    tau' = s (R0^T (t_h - t0) + dM c)     (per hypothesis; fp64)
// and because R0 is orthogonal, in exact arithmetic
//     s^2 |R_h p + t_h - q|^2 = |dM P' + tau' - V'|^2
//         = sum_ij (-2 dM_ij + G_ij) Q'_i P'_j + |V'|^2 + |tau'|^2 + 2 (dM^T tau') . P' - 2 tau' . V'  -  V'^T G P' ,
// G = R_h^T R_h - I the hypothesis' own defect (~1e-7 for a Kabsch result; the last term, <= 3 g |V'| |P'|, goes into the bound).
// That is a DOT PRODUCT of 16 features of the correspondence
//     Q'_i P'_j (9), |V'|^2 / 2, 256 P'_j (3), 256 V'_i (3)                         (tile workgroups of the Kabsch launch, fp64)
// with 16 coefficients of the hypothesis  -2 dM_ij + G_ij, 2, 2 (dM^T tau')_j / 256, -2 tau'_i / 256  plus the constant |tau'|^2
// (the Kabsch launch's own threads, fp64).  For a hypothesis NEAR the reference dM is small (C2: 0.02 - 0.05) and so is every
// term: the nine products that cancelled at the size of the CLOUDS in r03's form (|P'|^2 + |Q'|^2 against tau'^2) now cancel at
// dM times that, and a true correspondence has a small V' besides.  Features and coefficients are split into two fp16 halves; hi x hi,
// hi x lo and lo x hi products are kept: 48 slots = three chained v_mfma_f32_32x32x16_f16, rows = 32 hypotheses, columns = 32
// correspondences, accumulator initialised to |tau'|^2 - LO: a lane then holds x = D~ - LO for 16 hypotheses of one
// correspondence — sign bit into a shift register (1 instruction), shell test 0 <= x < HI - LO as an unsigned min over
// the 16 (0.5), nothing else: 1.6 vector instructions per test against the linear filter's 4.75.
//
// The cut.  reach_h = |dM|_F Pn + |tau'| bounds |dM P' + tau'| over the whole cloud, so a correspondence with
// |V'| > reach_h + tau' + dE_h + margin is an outlier of h by the triangle inequality — whatever the canonical chain's roundings
// (dE_h, below).  Correspondences are dealt into NEAR (|V'| <= th_v: first tile rows) and FAR (last rows), hypotheses into NEAR
// (reach_h + dE_h <= th_h = GX_KAPPA tau': first coefficient rows) and the rest (last rows) — both by one atomic per workgroup inside
// the Kabsch launch; th_v = th_h + 1.05 tau' + 1e-3.  A filter workgroup whose 256 rows are all NEAR hypotheses walks the NEAR
// units only.  On the BASELINE scenes ~90 % of the ranked hypotheses are near the voted reference and 11 - 20 % of the
// correspondences are near (the inliers and what lies around them): three quarters of the tests are never made.  The order inside
// each class is whatever the atomics give — the counts are integers, every permutation gives the same ones; hperm / pperm lead back.
//
// What the squared form costs is precision.  How the matrix
// pipe adds the 16 products and C is not in the ISA text.  tools/ubench/mfma_numerics.hip probes it (profiles/
// r03_ubench_mfma_numerics.txt): the final rounding is to nearest-even; fp16 sub-normals are NOT flushed; terms far below the
// largest one lose their low bits before the sum (a product 1.5 beside 2^24 arrives as 1; beside a C of 2^24 it arrives
// whole) — consistent with aligning all 17 terms to the largest exponent, products counted one binade up, and cutting them at
// a 2^-25 fraction of it, which reproduces two thirds of 3000 random cancelling dot products bit for bit; and over 15 000
// such dot products |hardware - exact| never passed 3.75 x 2^-24 of the LARGEST term (17 cuts of < 2^-24 of it each would
// allow 17).  The bound takes 18.5 x 2^-24 of the largest term per MFMA — the cut model's worst case, five times the worst
// seen.  This is a MEASURED model of gfx950's matrix pipe, not an ISA guarantee: every context probes its own pipe before the
// first call that would use this filter (gram_guard_kernel), and the parity suite compares EVERY count with the canonical
// kernel's at the BASELINE shapes, on adversarial scenes and on random degenerate clouds.  With F = |dM|_F, m = max |dM_ij|,
// Tn = |tau'|, g = max |G_ab|, and the correspondences a row is tested against bounded by |V'| <= vb, |Q'| <= Qb (NEAR rows:
// vb = th_v, Qb = Pn + th_v; the others: vb = Pn + Qn, Qb = Qn), u = 2^-24, D* the exact value:
//     terms        2 (m + g/2) Qb Pn | vb^2 | 2 F Tn Pn | 2 Tn vb | Tn^2 + 1.5 tau'^2 (the constant):  Mh = 1.05 x the largest
//     MFMA 1       18.5 u Mh                                                                    (GX_ACC)
//     MFMAs 2, 3   their terms are <= 2^-10 of MFMA 1's; their C is the running value: 18.5 u (|x| + 3e-3 S) each — the part
//                  proportional to x (2.2e-6) is carried by the factors (1 -+ 1e-5) of LO / HI, the rest is the 1e-8 S term
//     splits       x = hi + lo + rem, |rem| <= 2^-22 |x|;  w F - (wh Fh + wh Fl + wl Fh) = wl Fl + ... <= 3.01 x 2^-22 |w F|:
//                  7.2e-7 Sl, Sl = (2 F + 3 g) Qb Pn + 2 F Tn Pn + 2 Tn vb  (exact arithmetic, no model)          (GX_Q)
//     norm         the norm feature's coefficient 2 RS alpha has no low half, so of its three products only hi x hi and
//                  hi x lo exist: its two pieces leave 2^-22 of it, 2.4e-7 vb^2                                   (GX_NORM)
//     accumulator  initial value RS (|tau'|^2 - LO_h) rounded to fp32: u Mh (inside GX_ACC's margin)
//     defect       - V'^T G P': <= 3 g vb Pn  (beyond g = 1e-3: not a rotation, the hypothesis is recounted exactly)
//     frame        R0^T R0 = I to 3e-16, P' / Q' / dM / tau' evaluated in fp64: < 1e-9 in these units (inside the 1e-4)
//   sum: eps_h = 1.1e-6 Mh + 7.5e-7 Sl + 2.5e-7 vb^2 + 1e-8 S + 3 g vb Pn + 1e-4                                  (gram_eps)
// The canonical fp32 chain itself deviates from exact arithmetic: its residual VECTOR by <= dE = sqrt(3) 4 u s (qmax + 1.75 pmax
// + |t|max) (original, uncentred magnitudes), its square by 3 more roundings.  So, with st = s sqrt(tau2):
//     D~ <  LO_h = (st - dE)^2 (1 - 1e-5) - eps_h   =>  canonical inlier;     D~ >= HI_h = (st + dE)^2 (1 + 1e-5) + eps_h  =>  outlier;
// Every hypothesis keeps its OWN shell: the filter wave scales row h by alpha_h = the largest power of two <= min(16, widest shell
// of its 32 rows / the row's own width) — exact in fp16 / fp32 — so that "0 <= x < W" with one W per wave tests alpha_h RS (D~ - LO_h)
// against at least alpha_h RS (HI_h - LO_h).  Hypotheses fall into four classes, decided with the coefficients:
//     normal   eps_h <= 0.25 st^2: filtered;        far      |tau'| - F Pn - Vn >= 1.05 st + dE: no correspondence can be an inlier,
//     padding  beyond n_local: count 0;              recount  everything else (non-finite, not a rotation, shell too wide): its
//                                                             group of 8 rows goes to the exact pass wholesale.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split2(double x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)(float)x;
  lo = (_Float16)(float)(x - (double)(float)hi);
}

// the vote in a launch of its own (stage hook, sharded stage C, certified paths: wherever the counting pass did not carry it)
__global__ __launch_bounds__(4 * GX_VOTE) void gram_ref_kernel(GramRefJob job) {
  __shared__ float lds[GX_REF_LDS_WORDS];
  gram_ref_block(job, lds);
}

// exclusive rank of this thread among the threads of its 256-thread workgroup for which `flag` holds, and their number
// (two barriers; every thread of the workgroup must call it)
__device__ __forceinline__ uint32_t block_rank256(bool flag, uint32_t* __restrict__ s_w /* 4 words of LDS */, uint32_t* total) {
  const uint64_t b = __ballot(flag);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t pre = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < 4; w++) { const uint32_t v = s_w[w]; pre += w < wave ? v : 0u; tot += v; }
  __syncthreads();
  *total = tot;
  return pre + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// tile: per group of 32 rows 4 x 32 uint4 — for block k (0: hi halves, 1: lo halves): the 32 first halves (slots 0..7), then the 32
// second halves (slots 8..15), so that lane l reads uint4 number 64 k + l of the group (linear, conflict-free).
//   slots   0..8 Q'_i P'_j (index 3 i + j)   9 norm piece   10..12 256 P'_j   13..15 256 V'_i
//   MFMA 1: coefficient hi x block 0 (norm hi);   MFMA 2: coefficient hi x block 1 (norm lo);   MFMA 3: coefficient lo x block 0
// Row of correspondence m: NEAR ones from the front in the order their workgroups' atomics arrive, FAR ones from row n - 1 down;
// rows [n, rows) are sentinels.  pperm[row] = m.
// one read of the frame per workgroup (all threads call; one barrier)
__device__ __forceinline__ void load_frame(const GramFrame* __restrict__ fr, GramFrameRO* __restrict__ s_fr) {
  static_assert(sizeof(GramFrameRO) == 176 && offsetof(GramFrame, pmax_o) == 168, "GramFrameRO is the head of GramFrame");
  if (threadIdx.x < sizeof(GramFrameRO) / 8) reinterpret_cast<double*>(s_fr)[threadIdx.x] = reinterpret_cast<const double*>(fr)[threadIdx.x];
  __syncthreads();
}

__device__ void gram_tile_block(const float* __restrict__ planes, int n, int ld, const FilterTileJob& job, uint32_t block,
                                uint32_t blocks) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_base[2];
  __shared__ GramFrameRO s_fr;
  const uint32_t m = block * 256 + threadIdx.x;
  for (uint32_t z = m; z < job.zero_words; z += blocks * 256) job.zero[z] = 0u;
  GramFrame* __restrict__ fr = job.coef.frame;
  load_frame(fr, &s_fr);
  const bool real = m < (uint32_t)n;
  double P[3] = {0.0, 0.0, 0.0}, Q[3] = {0.0, 0.0, 0.0}, V[3] = {0.0, 0.0, 0.0}, vv = 0.0;
  if (real) {
    const double s = s_fr.s;
    double dq[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      P[c] = s * ((double)planes[(size_t)c * ld + m] - s_fr.c[c]);
      dq[c] = (double)planes[(size_t)(3 + c) * ld + m] - s_fr.t0[c];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Q[i] = s * ((s_fr.R0[i] * dq[0] + s_fr.R0[3 + i] * dq[1] + s_fr.R0[6 + i] * dq[2]) - s_fr.c[i]);  // (R0^T dq)_i
      V[i] = Q[i] - P[i];
      vv += V[i] * V[i];
    }
  }
  const bool near = real && vv <= s_fr.th_v * s_fr.th_v;
  uint32_t tot_near;
  const uint32_t r_near = block_rank256(near, s_w, &tot_near);
  // (the real correspondences of a workgroup are its first threads: rank among the FAR ones = thread - rank among the NEAR ones)
  const uint32_t n_real = (uint32_t)n > block * 256 ? min(256u, (uint32_t)n - block * 256) : 0u, tot_far = n_real - tot_near;
  const uint32_t r_far = threadIdx.x - r_near;
  if (threadIdx.x == 0) {  // ONE atomic per workgroup: both counts in one 64-bit word
    const unsigned long long was = n_real ? atomicAdd(&fr->pts, (unsigned long long)tot_near | ((unsigned long long)tot_far << 32)) : 0ull;
    s_base[0] = (uint32_t)was; s_base[1] = (uint32_t)(was >> 32);
  }
  _Float16 h[16], l[16], n2[2];  // (made while thread 0's atomic is on its way)
  if (real) {
    double F[16];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) F[3 * i + j] = Q[i] * P[j];
    const double N = 0.5 * vv;
#pragma unroll
    for (int c = 0; c < 3; c++) { F[10 + c] = 256.0 * P[c]; F[13 + c] = 256.0 * V[c]; }
    F[9] = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) split2(F[k], h[k], l[k]);
    n2[0] = (_Float16)(float)N;
    n2[1] = (_Float16)(float)(N - (double)(float)n2[0]);  // what is left: < 2^-22 N (GX_NORM)
  } else {  // sentinel: far away under every hypothesis (D~ = 2 x 60000 + ...), never undecided
#pragma unroll
    for (int k = 0; k < 16; k++) { h[k] = (_Float16)0.f; l[k] = (_Float16)0.f; }
    n2[0] = (_Float16)60000.f; n2[1] = (_Float16)0.f;
  }
  __syncthreads();
  if (m >= job.rows) return;
  const uint32_t row = near ? s_base[0] + r_near : (real ? (uint32_t)n - 1u - (s_base[1] + r_far) : m);
  if (real) job.coef.pperm[row] = m;
  uint4* tile = static_cast<uint4*>(job.tile) + (size_t)(row >> 5) * (32 * GX_TILE_Q) + (row & 31u);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const _Float16* v = k == 1 ? l : h;
    half8 f0 = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    half8 f1 = {v[8], n2[k], v[10], v[11], v[12], v[13], v[14], v[15]};
    tile[64 * k] = *reinterpret_cast<uint4*>(&f0);
    tile[64 * k + 32] = *reinterpret_cast<uint4*>(&f1);
  }
}

// Per-hypothesis coefficients of the Gram filter, made ONCE per call (by the Kabsch launch's own threads, or by
// gram_coef_kernel for the stage hook): thread = hypothesis l of this rank's shard, a whole 256-thread workgroup calls it.
// The row a hypothesis gets: NEAR ones from the front, everything else from row ldl - 1 down (one atomic per workgroup and
// class); hperm[row] = l.
//   coef.A      64 bytes per row: [hi halves of the 16 coefficients | lo halves] (NOT yet scaled by the row's alpha)
//   coef.C      RS (|tau'|^2 - LO_h), or +huge for a row that is switched off
//   coef.W      RS (HI_h - LO_h), 0 for a row that is not filtered
//   coef.flag   bit 0 normal (filtered), bit 1 recount (its group of 8 rows goes to the exact pass), bits 8.. log2 of the largest
//               alpha the row's coefficients can take without leaving fp16's range
__device__ void gram_coef_block(const GramCoef& coef, const float v[12], uint32_t l, uint32_t ldl, uint32_t n_local, float tau2) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_base[2];
  __shared__ GramFrameRO s_fr;
  GramFrame* __restrict__ fr = coef.frame;
  load_frame(fr, &s_fr);
  const double s = s_fr.s, st = s_fr.st, Pn = s_fr.Pn, Qn = s_fr.Qn, Vn = Pn + Qn;
  double R[9], dM[9], G[9], Tp[3];
#pragma unroll
  for (int c = 0; c < 9; c++) R[c] = (double)v[c];
  double F2 = 0.0, mm = 0.0, gdef = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      // (fp64 products of this block are fused by hand: the build runs -ffp-contract=off for the canonical fp32 chains, and an
      // unfused a b + c is two half-rate instructions here — C4's 500 000 hypotheses make this launch 12 % of its step)
      const double Mij = __builtin_fma(s_fr.R0[i], R[j], __builtin_fma(s_fr.R0[3 + i], R[3 + j], s_fr.R0[6 + i] * R[6 + j]));  // (R0^T R)_ij
      const double d = Mij - (i == j ? 1.0 : 0.0);
      dM[3 * i + j] = d;
      F2 = __builtin_fma(d, d, F2);
      mm = fmax(mm, fabs(d));
    }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {  // (R^T R - I)_ij, symmetric
      const double g = __builtin_fma(R[i], R[j], __builtin_fma(R[3 + i], R[3 + j], __builtin_fma(R[6 + i], R[6 + j], i == j ? -1.0 : 0.0)));
      G[3 * i + j] = g; G[3 * j + i] = g;
      gdef = fmax(gdef, fabs(g));
    }
  {
    const double d0 = (double)v[9] - s_fr.t0[0], d1 = (double)v[10] - s_fr.t0[1], d2 = (double)v[11] - s_fr.t0[2];
#pragma unroll
    for (int i = 0; i < 3; i++)
      Tp[i] = s * __builtin_fma(s_fr.R0[i], d0, __builtin_fma(s_fr.R0[3 + i], d1, __builtin_fma(s_fr.R0[6 + i], d2,
                   __builtin_fma(dM[3 * i], s_fr.c[0], __builtin_fma(dM[3 * i + 1], s_fr.c[1], dM[3 * i + 2] * s_fr.c[2])))));
  }
  float nanp = 0.f;
#pragma unroll
  for (int c = 0; c < 12; c++) nanp += v[c] * 0.f;
  // |dM|_F and |tau'|: fp32 square roots, pushed up / down by more than their rounding — every bound below is monotone in them
  const double T2 = __builtin_fma(Tp[0], Tp[0], __builtin_fma(Tp[1], Tp[1], Tp[2] * Tp[2]));
  const double Fn = (double)sqrt_rn((float)F2) * (1.0 + 4e-7) + 1e-30, Tr = (double)sqrt_rn((float)T2);
  const double Tn = Tr * (1.0 + 4e-7) + 1e-30, Tn_lo = Tr * (1.0 - 4e-7);
  const double tmax_o = fmax(fabs((double)v[9]), fmax(fabs((double)v[10]), fabs((double)v[11])));
  const double dE = GX_CANON * s * ((double)s_fr.qmax_o + 1.75 * (double)s_fr.pmax_o + tmax_o);
  const bool finite = nanp == 0.f;
  const bool pad = l >= n_local;
  const bool rot = finite && gdef <= 1e-3 && T2 < 1e12;
  const bool far = rot && (Tn_lo - Fn * Pn - Vn >= 1.05 * st + dE + 1e-3);
  // The hypothesis' own cut: a correspondence with |V'| > vb_h = reach_h + 1.05 st + dE_h is an outlier of h (triangle inequality),
  // so the shell only has to hold for |V'| <= vb_h — PROVIDED the filter cannot call such a correspondence an inlier either: with
  // U(v) the sum form of the bound, D~ >= (v - reach_h)^2 - U(v), which grows with v from vb_h on (its slope there is
  // 2.1 st - U' > 0 for st >= 0.2), so U(vb_h) <= 0.1 st^2 < (1.05^2 - 1) st^2 keeps D~ above st^2 >= LO_h.  Otherwise the shell is made
  // for every correspondence of the call (vb = Vn).
  const double reach = Fn * Pn + Tn;
  double vb = reach + 1.05 * st + dE + 1e-3, Qb = fmin(Qn, Pn + vb);
  const bool own = rot && vb < Vn && st >= 0.2 && gram_eps_sum(Fn, mm, Tn, gdef, Pn, Qb, vb, st) <= 0.1 * st * st;
  if (!own) { vb = Vn; Qb = Qn; }
  const double eps = gram_eps(Fn, mm, Tn, gdef, Pn, Qb, vb, st);
  // NEAR the reference: its workgroup of the filter looks only at the NEAR correspondences (|V'| <= th_v = th_h + 1.05 st + 1e-3)
  const bool close = own && !far && !pad && (reach + dE <= s_fr.th_h);
  // coefficients (A operand), scaled by GX_RS; the P' / V' features are stored x 256
  double a16[16], cmax = 0.0;
#pragma unroll
  for (int k = 0; k < 9; k++) a16[k] = (double)GX_RS * __builtin_fma(-2.0, dM[k], G[k]);
  a16[9] = 2.0 * (double)GX_RS;
#pragma unroll
  for (int j = 0; j < 3; j++) a16[10 + j] = 2.0 * (double)GX_RS / 256.0 * __builtin_fma(dM[j], Tp[0], __builtin_fma(dM[3 + j], Tp[1], dM[6 + j] * Tp[2]));
#pragma unroll
  for (int i = 0; i < 3; i++) a16[13 + i] = -2.0 * (double)GX_RS / 256.0 * Tp[i];
#pragma unroll
  for (int k = 0; k < 16; k++) cmax = fmax(cmax, fabs(a16[k]));
  uint32_t lg = 4;  // alpha <= 16
  while (lg > 0 && cmax * (double)(1u << lg) > 30000.0) lg--;
  // (st <= 32: the sentinel correspondences' 2 x 60000 must stay far above every threshold)
  const bool normal = rot && !far && !pad && (eps <= 0.25 * st * st) && (dE <= 0.1 * st) && (st <= 32.0) && (cmax <= 30000.0);
  const bool recount = !pad && !far && !normal;
  const bool good = close && normal;
  const double LOh = (st - dE) * (st - dE) * (1.0 - 1e-5) - eps, HIh = (st + dE) * (st + dE) * (1.0 + 1e-5) + eps;
  uint32_t n_good;
  const uint32_t r_good = block_rank256(good, s_w, &n_good), r_bad = threadIdx.x - r_good, n_bad = 256u - n_good;
  // this workgroup's segment of the coefficient rows (sc_gramref.hpp): NEAR hypotheses from its front, the others from its back
  const uint32_t groups = ldl / 256u, S = gram_segments(groups), wg = l / 256u, sg = wg % S, seg_rows = 256u * ((groups - sg + S - 1u) / S);
  if (threadIdx.x == 0) {  // ONE atomic per workgroup: both counts in one 64-bit word
    const unsigned long long was = atomicAdd(&fr->seg[sg].cnt, (unsigned long long)n_good | ((unsigned long long)n_bad << 32));
    s_base[0] = (uint32_t)was; s_base[1] = (uint32_t)(was >> 32);
  }
  // (the fp16 halves are made while thread 0's atomic is on its way: ~1.5 us of latency every workgroup used to sit out)
  _Float16 ah[16], al[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    split2(a16[k], ah[k], al[k]);
    if (!normal) { ah[k] = (_Float16)0.f; al[k] = (_Float16)0.f; }
  }
  al[9] = (_Float16)0.f;  // (2 RS is a power of two: no low half)
  const float c_row = normal ? (float)((double)GX_RS * (T2 - LOh)) : 1e30f;
  const float w_row = normal ? (float)((double)GX_RS * (HIh - LOh) * (1.0 + 1e-6)) : 0.0f;
  __syncthreads();
  if (l >= ldl) return;
  const uint32_t row = gram_seg_row(good ? s_base[0] + r_good : seg_rows - 1u - (s_base[1] + r_bad), sg, S);
  coef.hperm[row] = l;
  coef.C[row] = c_row;
  coef.W[row] = w_row;
  coef.flag[row] = (normal ? 1u : 0u) | (recount ? 2u : 0u) | (lg << 8);
  uint4* __restrict__ out = reinterpret_cast<uint4*>(coef.A) + (size_t)row * 4;
  half8 q0 = {ah[0], ah[1], ah[2], ah[3], ah[4], ah[5], ah[6], ah[7]}, q1 = {ah[8], ah[9], ah[10], ah[11], ah[12], ah[13], ah[14], ah[15]};
  half8 q2 = {al[0], al[1], al[2], al[3], al[4], al[5], al[6], al[7]}, q3 = {al[8], al[9], al[10], al[11], al[12], al[13], al[14], al[15]};
  out[0] = *reinterpret_cast<uint4*>(&q0); out[1] = *reinterpret_cast<uint4*>(&q1);
  out[2] = *reinterpret_cast<uint4*>(&q2); out[3] = *reinterpret_cast<uint4*>(&q3);
}

__global__ __launch_bounds__(256) void gram_coef_kernel(const float* __restrict__ RtSoA, uint32_t ldl, uint32_t n_local, float tau2,
                                                        GramCoef coef) {
  const uint32_t l = blockIdx.x * 256 + threadIdx.x;
  float v[12];
#pragma unroll
  for (int c = 0; c < 12; c++) v[c] = l < ldl ? RtSoA[(size_t)c * ldl + l] : 0.f;
  gram_coef_block(coef, v, l, ldl, n_local, tau2);
}

// ------------------------------------------------------------------------------------------------
// Run-time guard of the Gram filter's error model.  GX_ACC — "one v_mfma_f32_32x32x16_f16 is off by at most 18.5 x 2^-24 of
// its LARGEST term" — is a measured property of gfx950's matrix pipe (tools/ubench/mfma_numerics.hip), not an ISA guarantee,
// and the Gram filter's shells are only as wide as it says.  So every context probes the pipe it actually runs on, once,
// before the first call that would choose the Gram filter: GUARD_WAVES x GUARD_ITERS MFMAs of pseudo-random, strongly
// cancelling dot products (the microbenchmark's recipe: 1 .. 16 non-zero slots, factors spread over 12 / 14 binades, C = minus
// the exact sum within 10 % plus noise, or zero), each of the 1024 results of an instruction compared with the fp64 value of
// the same sum — fp16 x fp16 products are exact in fp64 and seventeen of them add with ~2^-49 of error, nothing against
// 2^-24.  The probe reports the largest |hardware - exact| / largest |term| it saw, in units of 2^-24, and whether fp16
// sub-normal operands were kept (the low halves of small features are sub-normal: a pipe that flushes them breaks GX_Q).
// A context whose pipe exceeds GUARD_LIMIT (half of what the bound allows; this silicon shows ~4) or flushes sub-normals
// never runs the Gram filter: its calls take the linear filter, whose bound allows a truncating accumulation.
// ------------------------------------------------------------------------------------------------
constexpr int GUARD_WAVES = 64, GUARD_ITERS = 16;
constexpr float GUARD_LIMIT = 9.25f;  // x 2^-24 of the largest term: GX_ACC / 2

__device__ __forceinline__ uint64_t guard_hash(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// operand (which = 0: A row / 1: B column) `rc`, slot k of MFMA number `id`: an fp16 value any lane can re-derive
__device__ __forceinline__ float guard_operand(uint32_t id, int which, int rc, int k, int nz) {
  if (k >= nz) return 0.f;
  const uint64_t hsh = guard_hash(((uint64_t)id << 20) | ((uint64_t)which << 16) | ((uint64_t)rc << 8) | (uint64_t)k);
  const float u = (float)(hsh & 0xFFFFFFu) * (1.0f / 16777216.0f) - 0.5f;
  const int e = (int)((hsh >> 24) % (which ? 14u : 12u));
  return (float)(_Float16)ldexpf(u, e);
}

__global__ __launch_bounds__(64) void gram_guard_kernel(uint32_t* __restrict__ out) {
  // out[0]: max over all results of |hw - exact| / largest term (fp32 bits of a non-negative value: atomicMax orders them);
  // out[1]: sub-normal operands kept (1 set by the wave that tests it); out[2]: results compared
  const int lane = threadIdx.x, rc = lane & 31, hf = lane >> 5;
  float worst = 0.f;
  for (int it = 0; it < GUARD_ITERS; it++) {
    const uint32_t id = blockIdx.x * GUARD_ITERS + it;
    const int nz = 1 + (int)(guard_hash(0xABCD0000ull + id) % 16u);
    half8 A, B;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      A[e] = (_Float16)guard_operand(id, 0, rc, 8 * hf + e, nz);
      B[e] = (_Float16)guard_operand(id, 1, rc, 8 * hf + e, nz);
    }
    // lane (col = rc, hf) receives D[i] for rows 8 (i >> 2) + 4 hf + (i & 3): it re-derives those rows' operands for the exact sums
    double bcol[16];
#pragma unroll
    for (int k = 0; k < 16; k++) bcol[k] = (double)guard_operand(id, 1, rc, k, nz);
    f32x16 C;
    double want[16], big[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = 8 * (i >> 2) + 4 * hf + (i & 3);
      double ex = 0.0, mx = 0.0;
      for (int k = 0; k < 16; k++) {
        const double t = (double)guard_operand(id, 0, r, k, nz) * bcol[k];
        ex += t;
        mx = fmax(mx, fabs(t));
      }
      const uint64_t hc = guard_hash(0x5EED000000ull + ((uint64_t)id << 12) + (uint64_t)(r * 32 + rc));
      const double u1 = (double)(hc & 0xFFFFFu) / 1048576.0, u2 = (double)((hc >> 20) & 0xFFFFFu) / 1048576.0;
      const float c = ((hc >> 40) % 3u == 0u) ? 0.f : (float)(-ex * (0.9 + 0.2 * u1) + (u2 - 0.5) * 64.0);
      C[i] = c;
      want[i] = ex + (double)c;
      big[i] = fmax(mx, fabs((double)c));
    }
    const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (big[i] > 0.0) worst = fmaxf(worst, (float)(fabs((double)D[i] - want[i]) / big[i] * 16777216.0));
  }
  if (!(worst >= 0.f)) worst = 1e30f;  // NaN: a violation
  atomicMax(&out[0], __float_as_uint(worst));
  if (lane == 0) atomicAdd(&out[2], (uint32_t)(GUARD_ITERS * 1024));
  if (blockIdx.x == 0) {  // sub-normal operands: 2^-20 x 1024 on either side must arrive as 2^-10
    half8 A, B;
#pragma unroll
    for (int e = 0; e < 8; e++) { A[e] = (_Float16)0.f; B[e] = (_Float16)0.f; }
    if (hf == 0) {  // slot 0 only: rows 0 .. 15 of A hold the sub-normal 2^-20, rows 16 .. 31 hold 1024; B holds 1024
      A[0] = (_Float16)(rc < 16 ? 9.5367431640625e-07f : 1024.f);
      B[0] = (_Float16)1024.f;
    }
    // second product: the sub-normal on the B side (columns 16 .. 31), met by the rows of A that hold 1024
    half8 B2 = B;
    if (hf == 0 && rc >= 16) B2[0] = (_Float16)9.5367431640625e-07f;
    f32x16 Z;
#pragma unroll
    for (int i = 0; i < 16; i++) Z[i] = 0.f;
    const f32x16 D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, Z, 0, 0, 0);   // rows 0..15: 2^-20 x 1024
    const f32x16 D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B2, Z, 0, 0, 0);  // columns 16..31 x rows 16..31: 1024 x 2^-20
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = 8 * (i >> 2) + 4 * hf + (i & 3);
      if (r < 16) ok = ok && D1[i] == 0.0009765625f;
      if (r >= 16 && rc >= 16) ok = ok && D2[i] == 0.0009765625f;
    }
    const uint64_t all = __ballot(ok);
    if (lane == 0) out[1] = (all == ~0ull) ? 1u : 0u;
  }
}

hipError_t gram_guard_probe(void* scratch, hipStream_t st, float* worst_units, bool* subnormals_kept, uint32_t* compared) {
  uint32_t h[4] = {0, 0, 0, 0};
  hipError_t e = hipMemsetAsync(scratch, 0, 16, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gram_guard_kernel, dim3(GUARD_WAVES), dim3(64), 0, st, static_cast<uint32_t*>(scratch));
  e = hipMemcpyAsync(h, scratch, 16, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  union { uint32_t u; float f; } w; w.u = h[0];
  *worst_units = w.f;
  *subnormals_kept = h[1] == 1u;
  if (compared) *compared = h[2];
  return hipSuccess;
}
float gram_guard_limit() { return GUARD_LIMIT; }
bool filter_ablations_built() {
#ifdef SC_ABLATIONS
  return true;
#else
  return false;
#endif
}

// sum of `v` over each aligned group of 32 lanes, valid in the group's upper 16 lanes (DPP: quad swaps, half mirror, mirror,
// then lane 15 of the lower row broadcast into the upper row)
__device__ __forceinline__ uint32_t dpp_sum32_upper(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror: every lane holds its row's sum
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast15 into rows 1 and 3
  return v;
}

template <int VAR>  // (timing-only ablations for tools/ab_stage.py: 512 = no shell test, 256 = no barrier, 32 = no epilogue)
__global__ __launch_bounds__(64 * GX_WAVES, 2) void score_gram_kernel(GramCoef coef, uint32_t ldl,
                                                                     const uint4* __restrict__ tile, uint32_t windows,
                                                                     uint32_t splits, uint32_t n_waves8,
                                                                     uint32_t* __restrict__ cnt_out, uint2* __restrict__ gq,
                                                                     uint32_t cap_sq, uint32_t* __restrict__ qcount,
                                                                     uint32_t* __restrict__ redo_bits, uint32_t ql, uint32_t group_major) {
  __shared__ uint4 Bt[3][GX_UNIT * GX_TILE_Q];   // ring of three 16 KiB units: one being read, one landing, one being issued
  __shared__ uint2 queue[GX_WAVES][GX_QL];
  constexpr int PIECES = GX_UNIT * GX_TILE_Q / (64 * GX_WAVES);  // LDS-DMA instructions per wave and unit
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hf = lane >> 5;
#ifdef SC_ABLATIONS  // lab build: where a workgroup's life goes (tools/r4/gramtick.py; ql == 127 switches the printing on)
  long long tk[5];
  tk[0] = wall_clock64();
#define SC_GRAM_TICK(i) tk[i] = wall_clock64()
#else
#define SC_GRAM_TICK(i)
#endif
  // The grid is one-dimensional, groups x splits workgroups, and the LONG ones go first: the row blocks that hold
  // a hypothesis not near the reference are the LAST ones of their segment (sc_gramref.hpp: segment = row block % S) — the highest
  // row blocks — and walk every correspondence, the others a few near units (dispatched in row order, the last workgroups to
  // start were the longest: a tail as long as the kernel's useful part).
  const GramFrame* __restrict__ fr = coef.frame;
  const uint32_t groups = gridDim.x / splits, S = gram_segments(groups);
  // GROUP-major, highest row blocks first (r04c): EVERY split of a far row block is dispatched before the first near one.  Split-
  // major (r04b) started split k's far workgroups — the longest of the launch — only after the splits before it had been
  // dispatched, near workgroups, empty ones and all (C4: the second split's 267 far workgroups of 10 units began ~45 us into the
  // launch).  Within a row block the splits are rotated by r(block): workgroup b runs on XCD b % 8, and with by = b % splits the
  // WORKING workgroups of the near rows (by < ns) would sit on 8 / gcd(splits, 8) of the eight XCDs (by = b % 8 at 8 splits put
  // them all on one: C3 508 us instead of 150); r advances once per 8 / gcd blocks, which deals them to all eight evenly.
  // Only for launches of about two resident generations (launch_score_filter: C2's 980 workgroups, 22.6 -> 20.2 us): at C4's 3908
  // the same order measured 147 us against 109 — two far workgroups sharing a CU run at half speed each, and far-first pairs them
  // all; split-major pairs most of them with short near ones.  There: split-major, the highest row blocks first within a split.
  uint32_t bx, by;
  if (group_major) {
    const uint32_t gidx = blockIdx.x / splits, sidx = blockIdx.x - gidx * splits;
    const uint32_t dg = min(splits & (0u - splits), 8u);            // gcd(splits, 8): splits <= 8
    bx = groups - 1u - gidx; by = (sidx + ((gidx * dg) >> 3) % dg) % splits;
  } else {
    bx = groups - 1u - blockIdx.x % groups; by = blockIdx.x / groups;
  }
  const uint32_t wid = bx * GX_WAVES + wave;  // wave of 32 hypotheses
  // The cut: a workgroup whose 256 rows are all hypotheses NEAR the reference walks the NEAR correspondences only — the tile's
  // first rows; every other correspondence is an outlier of every one of them by the triangle inequality (see the header).
  const GramGeom gg = gram_geom(windows, splits, groups, fr);
  const bool near_block = gram_near_block(bx, S, fr);
  if (near_block && by >= gg.ns) return;  // (its rows of these splits are cleared by split 0, below)
  const uint32_t u0 = by * (near_block ? gg.pu_near : gg.pu_far);
  const uint32_t u1 = near_block ? min(u0 + gg.pu_near, gg.near_units) : min(u0 + gg.pu_far, gg.units);
  // ---- every global read of the prologue is ISSUED here, before anything waits: the coefficients of this lane's row, the
  // per-row words of the wave's 16 accumulator rows.  (Written where they are used, each group of four rows became a load ->
  // wait -> branch chain of its own, and the epilogue sixteen load -> wait -> store round trips: by the lab build's timestamps
  // a workgroup of C2 spent 4 - 5.7 us before its first step and 3.1 - 4.1 us after its last one, as long as in the 4.6 - 7.9 us
  // of steps between them.)  Behind the early exit, not before it: issued first of all — under the frame's own two dependent
  // reads — the workgroups with nothing to do, most of C2's grid, read ~230 bytes per thread for nothing: 24.1 -> 26.2 us.
  const uint32_t row = (uint32_t)col, h = wid * 32 + row;  // (< ldl: a multiple of 256, the grid has ldl / 256 workgroups)
  const uint4* __restrict__ ca = reinterpret_cast<const uint4*>(coef.A) + (size_t)h * 4;
  typedef float f32q __attribute__((ext_vector_type(4)));
  typedef uint32_t u32q __attribute__((ext_vector_type(4)));
  const u32q a0 = *reinterpret_cast<const u32q*>(ca + hf), a2 = *reinterpret_cast<const u32q*>(ca + 2 + hf);
  const float wrow = coef.W[h];
  const uint32_t frow = coef.flag[h], hrow = coef.hperm[h];
  f32q c4[4], w4[4];
  u32q f4[4];
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const size_t base = (size_t)wid * 32 + 8 * jj + 4 * hf;
    c4[jj] = *reinterpret_cast<const f32q*>(coef.C + base);
    w4[jj] = *reinterpret_cast<const f32q*>(coef.W + base);
    f4[jj] = *reinterpret_cast<const u32q*>(coef.flag + base);
  }
  // the places in the ranked list of this lane's 16 accumulator rows (read at the end, 16 registers the step loop has no room for)
  auto ranked_rows = [&](uint32_t* hp) {
    u32q hp4[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) hp4[jj] = *reinterpret_cast<const u32q*>(coef.hperm + (size_t)wid * 32 + 8 * jj + 4 * hf);
    asm volatile("" ::"v"(hp4[0]), "v"(hp4[1]), "v"(hp4[2]), "v"(hp4[3]));  // (four loads, ONE wait)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) { hp[4 * jj] = hp4[jj].x; hp[4 * jj + 1] = hp4[jj].y; hp[4 * jj + 2] = hp4[jj].z; hp[4 * jj + 3] = hp4[jj].w; }
  };
  auto stage = [&](uint32_t u, int buf) {  // (asm: see score_filter_kernel)
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
      const uint4* gsrc = tile + (size_t)u * (GX_UNIT * GX_TILE_Q) + 64 * GX_WAVES * i + tid;
      const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&Bt[buf][64 * GX_WAVES * i + wave * 64]));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  if (u0 < u1) stage(u0, 0);
  if (u0 + 1 < u1) stage(u0 + 1, 1);
  // (the asm pins the loads above the branches below: nothing is sunk into the block that uses it)
  asm volatile("" ::"v"(a0), "v"(a2), "v"(wrow), "v"(frow), "v"(c4[0]), "v"(c4[1]), "v"(c4[2]), "v"(c4[3]), "v"(w4[0]), "v"(w4[1]),
               "v"(w4[2]), "v"(w4[3]));
  asm volatile("" ::"v"(f4[0]), "v"(f4[1]), "v"(f4[2]), "v"(f4[3]), "v"(hrow));
  if (near_block && by == 0 && hf == 0 && h < ldl)  // the splits no workgroup of these rows works in count nothing
    for (uint32_t y = gg.ns; y < splits; y++) cnt_out[(size_t)y * ldl + hrow] = 0u;
  if (u0 >= u1) {  // (workgroup-uniform) nothing to look at in this split
    if (hf == 0 && h < ldl) cnt_out[(size_t)by * ldl + hrow] = 0u;
    return;
  }
  // ---- prologue: the coefficients were made once per call by gram_coef_block (lane (row, hf) takes the halves of its row).  What
  // belongs to the WAVE — the widest shell of its 32 rows, every row's alpha, the groups of 8 rows the exact pass recounts — is
  // made here: the rows of a wave come from all over the ranked list (NEAR hypotheses first), only this kernel sees them together.
  half8 A0 = *reinterpret_cast<const half8*>(&a0), A2 = *reinterpret_cast<const half8*>(&a2);
  float wmax = wrow;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 32));  // (both lane halves hold the same 32 rows)
  const uint32_t normal_rows = (uint32_t)__ballot((frow & 1u) != 0u), rc = (uint32_t)__ballot((frow & 2u) != 0u);
  uint32_t redo4 = 0, live_rows = normal_rows;
#pragma unroll
  for (int jj = 0; jj < 4; jj++)
    if ((rc >> (8 * jj)) & 0xFFu) { redo4 |= 1u << jj; live_rows &= ~(0xFFu << (8 * jj)); }  // rows of a group that is recounted anyway are switched off
  const bool any_normal = live_rows != 0u;
  // alpha: the largest power of two <= min(what the row's fp16 coefficients allow, widest shell / own shell) — scaling by it is exact
  auto alpha_of = [&](float w, uint32_t fl) -> float {  // (no division: w 2^k is exact, the comparisons decide; 17 correctly rounded divides per lane cost the prologue ~350 instructions)
    const int cap = (int)((fl >> 8) & 7u);
    float a = 1.0f;
#pragma unroll
    for (int k = 1; k <= 4; k++) a = (k <= cap && w * (float)(1 << k) <= wmax) ? (float)(1 << k) : a;
    return a;
  };
  {
    const bool live = (live_rows >> row) & 1u;
    const _Float16 am = (_Float16)(live ? alpha_of(wrow, frow) : 0.0f);
    A0 = A0 * am; A2 = A2 * am;
  }
  const half8 A1 = A0;  // hi x lo: the tile's second block holds the low halves of the features
  // alpha_h x own width <= wmax: one W for the wave (a hair above wmax)
  const uint32_t W2b = ((VAR & 512) || !any_normal) ? 0u : __float_as_uint(wmax * (1.0f + 4e-7f));
  f32x16 C;
#pragma unroll
  for (int jj = 0; jj < 4; jj++) {
    const float cc[4] = {c4[jj].x, c4[jj].y, c4[jj].z, c4[jj].w}, ww[4] = {w4[jj].x, w4[jj].y, w4[jj].z, w4[jj].w};
    const uint32_t ff[4] = {f4[jj].x, f4[jj].y, f4[jj].z, f4[jj].w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool live = (live_rows >> (8 * jj + 4 * hf + i)) & 1u;
      C[4 * jj + i] = live ? cc[i] * alpha_of(ww[i], ff[i]) : 1e30f;
    }
  }
  // The coefficients are USED here, so that the compiler's wait for these global loads sits here and not at their first real
  // use — the first MFMA of the step loop, where an s_waitcnt vmcnt(0) in every trip also waited for the LDS-DMA of the units
  // ahead (the compiler cannot see into the asm that issues them): until r03c every unit's first step stood still until the
  // NEXT unit had landed, i.e. nothing was prefetched at all.
  asm volatile("" ::"v"(A0), "v"(A2), "v"(C), "s"(W2b), "s"(redo4));
  SC_GRAM_TICK(1);
  uint32_t total[16], sr[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { total[i] = 0; sr[i] = 0; }
  const int my_step = __builtin_amdgcn_readfirstlane(wave) & (GX_UNIT / 32 - 1);  // the step at which this wave issues its LDS-DMA
  uint32_t qn = 0;  // entries in this wave's queue: wave-uniform, lives in a scalar register
  uint2* q = queue[wave];
  bool flushed = false;  // a flush's global stores are younger than the LDS-DMA pieces in flight: the next wait is a full one
  auto flush_queue = [&]() {  // the wave's queue -> its global sub-queue (one ticket); see score_filter_kernel
    const uint32_t sq = wid % FX_NQ;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&qcount[sq * 32], qn);
    base = __shfl(base, 0);
    const bool fits = base + qn <= cap_sq;
    for (uint32_t i = lane; i < qn && base + i < cap_sq; i += 64)
      gq[(size_t)sq * cap_sq + base + i] = fits ? make_uint2(q[i].x, (wid << 17) | q[i].y) : make_uint2(0u, 0u);
    if (!fits) redo4 = 0xFu;
    qn = 0;
    flushed = true;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the entries were read before the queue is written again
  };
  for (uint32_t u = u0; u < u1; u++) {
    const int buf = (int)((u - u0) % 3u);
    if constexpr ((VAR & 256) == 0) {
      // this wave's part of unit u has landed: everything it issued except the youngest unit's pieces (loads return in order)
      if (u + 1 < u1 && !flushed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flushed = false;
      __syncthreads();  // ... everybody's has, and nobody reads unit u - 1's buffer any more (it takes unit u + 2)
    }
#ifdef SC_ABLATIONS
    if (u == u0) tk[2] = wall_clock64();
#endif
    // Unit u + 2 is issued right behind the barrier, TWO units ahead: with two buffers the pieces of unit u + 1 were issued here
    // and — see the note on the coefficients above — waited for at once.  (VAR bit 0: each wave issues at a step of its own
    // inside the step loop instead, so that the eight waves' LDS-DMA instructions, ~125 cycles apiece, do not queue up behind
    // each other — tools/ubench/gram_loop.hip loses 20 % to that; here it measured 2 % SLOWER, C4 319 vs 313 us.)
    const bool stage_ahead = (VAR & 64) == 0 && u + 2 < u1;
    const int nbuf = buf == 0 ? 2 : buf - 1;
    if constexpr ((VAR & 1) == 0) {
      if (stage_ahead) stage(u + 2, nbuf);
    }
    if (any_normal && redo4 != 0xFu) {
      const half8* Bc = reinterpret_cast<const half8*>(Bt[buf]) + lane;
      // the vector work of one step.  sign bit: D~ < LO_h, a certain inlier.  Shell test 0 <= x < W as an unsigned minimum,
      // kept per group of four registers: with 1024 tests per step a step has an undecided test one time in four, so the
      // path that queues them must be short — it looks at the four group minima first and only at a group that hit.
      auto epilogue = [&](const f32x16& D, int g) {
        if constexpr ((VAR & 32) != 0) { sr[0] ^= __float_as_uint(D[0]) ^ __float_as_uint(D[5]) ^ __float_as_uint(D[10]) ^ __float_as_uint(D[15]); return; }
        uint32_t gm[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int i = 4 * j; i < 4 * j + 4; i++) sr[i] = __builtin_amdgcn_alignbit(sr[i], __float_as_uint(D[i]), 31);
          gm[j] = min(min(min(__float_as_uint(D[4 * j]), __float_as_uint(D[4 * j + 1])), __float_as_uint(D[4 * j + 2])), __float_as_uint(D[4 * j + 3]));  // v_min3 + v_min
        }
        const uint32_t mn = min(min(min(gm[0], gm[1]), gm[2]), gm[3]);
        if (__builtin_expect(__ballot(mn < W2b) != 0, 0)) {
          // A step has an undecided test one time in three, and a wave that dawdles here holds up the seven others at the next
          // barrier: nothing on this path waits for anything.  Groups of four registers that no lane hit are skipped (wave-
          // uniform branch); in a group that hit, the lanes take consecutive slots of the wave's LDS queue by the ballot mask
          // (v_mbcnt) above a fill count that lives in a scalar register — the LDS atomic with its returned value and the
          // s_waitcnt behind it that did this before cost a third of the kernel (C4: 341 us, 228 without this path).
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint64_t hit = __ballot(gm[j] < W2b);
            if (hit == 0) continue;
            const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
            if (gm[j] < W2b) {
              uint32_t bits = 0;
#pragma unroll
              for (int i = 0; i < 4; i++) bits |= (__float_as_uint(D[4 * j + i]) < W2b) ? (1u << (4 * j + i)) : 0u;
              if (slot < ql) q[slot] = make_uint2(u * GX_UNIT + 32 * g + col, ((uint32_t)hf << 16) | bits);
            }
            qn += (uint32_t)__popcll(hit);
          }
        }
      };
      // One step after the other: issuing step g + 1's chain before step g's vector work (a software pipeline with two
      // accumulator sets) was built and measured — hipcc rotates the loop into THREE sets with 16 moves per trip and
      // spills in the queueing path: C4 335 -> 403 us, C2 52 -> 57.  The waves of a SIMD overlap each other instead.
      half8 b0 = Bc[0], b1 = Bc[64];
#pragma unroll 1
      for (int g = 0; g < GX_UNIT / 32; g++) {
        f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0, C, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, b1, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, b0, D, 0, 0, 0);  // lo x hi: the hi halves of the features again
        // the next step's operands: two 16-byte LDS reads, in flight under this step's vector work
        if (g + 1 < GX_UNIT / 32) { b0 = Bc[32 * GX_TILE_Q * (g + 1)]; b1 = Bc[32 * GX_TILE_Q * (g + 1) + 64]; }
        if constexpr ((VAR & 1) != 0) {
          if (stage_ahead && g == my_step) stage(u + 2, nbuf);
        }
        epilogue(D, g);
      }
      // beyond ql entries were dropped — the exact pass takes the whole (wave, split)
      if (qn > ql) { redo4 = 0xFu; qn = 0; }
      if (((u - u0 + 1) & 3u) == 0) {  // every fourth unit: 32 tests per register
#pragma unroll
        for (int i = 0; i < 16; i++) { total[i] += (uint32_t)__popc(sr[i]); sr[i] = 0; }
      }
      if (qn > ql / 2) flush_queue();
    } else if ((VAR & 1) != 0 && stage_ahead) {
      stage(u + 2, nbuf);  // (a wave with nothing to filter still brings its share of the tile)
    }
  }
  SC_GRAM_TICK(3);
  if (qn && redo4 != 0xFu) flush_queue();
  uint32_t hp[16];
  ranked_rows(hp);
  if (lane == 0) {
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
      if (((redo4 >> jj) & 1u) && (wid * 4 + jj) < n_waves8) {
        const size_t bit = (size_t)by * n_waves8 + wid * 4 + jj;
        atomicOr(&redo_bits[bit >> 5], 1u << (bit & 31));
      }
  }
  // the 32 columns of a lane half are two DPP rows: five v_add_dpp per register instead of five dependent ds_bpermute round
  // trips (80 of them in a row were ~2.5 us at the end of every wave — a fifth of a wave's life at C2's 32 steps)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint32_t c = dpp_sum32_upper(total[i] + (uint32_t)__popc(sr[i]));  // (+ what a NEAR range that ends inside a window left in the shift register)
    const uint32_t r = 8 * (i >> 2) + 4 * hf + (i & 3), hh = wid * 32 + r;
    if (col == 31 && hh < ldl) cnt_out[(size_t)by * ldl + hp[i]] = ((redo4 >> (i >> 2)) & 1u) ? 0u : c;
  }
#ifdef SC_ABLATIONS
  SC_GRAM_TICK(4);
  if (ql == 127 && tid == 0 && blockIdx.x % 97 == 0)  // (wall_clock64: 100 MHz)
    printf("gram wg %u near %d units %u: prologue %lld, first unit %lld, steps %lld, tail %lld (x 10 ns)\n", blockIdx.x, (int)near_block,
           u1 - u0, tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3]);
#endif
#undef SC_GRAM_TICK
}

// (A persistent one-generation form of this kernel — contiguous runs of (hypothesis group, unit) items per workgroup, counts by
// atomics, recounts per unit: what VERDICT r03 #4 proposed — was built in round 4, bit-exact, and measured 3 - 4 % SLOWER at C2 and
// C4: profiles/r04_ab_gram_persistent.txt, DESIGN.md 5.0.  Its code left the tree when the reference-frame form below arrived.)

GramCoef gram_coef_view(void* buf, uint32_t ld_local, void* frame) {
  GramCoef g;
  unsigned char* p = static_cast<unsigned char*>(buf);
  g.A = p; p += (size_t)ld_local * 64;
  g.C = reinterpret_cast<float*>(p); p += (size_t)ld_local * 4;
  g.W = reinterpret_cast<float*>(p); p += (size_t)ld_local * 4;
  g.flag = reinterpret_cast<uint32_t*>(p); p += (size_t)ld_local * 4;
  g.hperm = reinterpret_cast<uint32_t*>(p); p += (size_t)ld_local * 4;
  g.pperm = reinterpret_cast<uint32_t*>(p);
  g.frame = static_cast<GramFrame*>(frame);
  return g;
}

FilterTileJob filter_tile_job(const FilterPlan& fp, const uint32_t* mx_cur, uint32_t* mx_next, void* tile, void* state,
                              void* coef, uint32_t ld_local, float tau2, void* frame) {
  const FilterState f = filter_state(state, fp);
  FilterTileJob j{fp.rows, mx_cur, mx_next, tile, f.info, f.qcount, f.zero_words, fp.mode,
                  GramCoef{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, tau2};
  if (fp.mode == 2) j.coef = gram_coef_view(coef, ld_local, frame);
  return j;
}

size_t gram_frame_bytes() { return sizeof(GramFrame); }  // (the class counters sit on 128-byte lines of their own)

GramRefJob gram_ref_job(const Points& pts, const uint32_t* mx, float tau2, const Tuning& tn, void* frame) {
  GramRefJob j{};
  j.planes = pts.planes; j.n = pts.n; j.ld = pts.ld;
  j.mx = mx; j.tau2 = tau2;
  j.kappa = tn.gram_kappa_q4 ? (float)tn.gram_kappa_q4 / 16.0f : (float)GX_KAPPA;
  j.out = static_cast<GramFrame*>(frame);
  return j;
}

// the frame of this call's Gram filter in a launch of its own: before the tile / the coefficients are made (it also clears their
// counters).  RtSoA: the hypotheses, if they exist already (stage hook); else they are solved from the selection `ts`.
void launch_gram_ref(const Points& pts, const TriSource& ts, const Shard& sh, const float* RtSoA, const uint64_t* t_eff_dev,
                     const uint32_t* mx, float tau2, const Tuning& tn, void* frame, hipStream_t st) {
  if (sh.n_local == 0) return;
  GramRefJob j = gram_ref_job(pts, mx, tau2, tn, frame);
  j.src.RtSoA = RtSoA; j.src.ts = ts; j.src.sh = sh; j.src.t_eff_dev = t_eff_dev;
  hipLaunchKernelGGL(gram_ref_kernel, dim3(1), dim3(4 * GX_VOTE), 0, st, j);
}

void launch_gram_coef(const float* RtSoA, const Shard& sh, float tau2, const GramCoef& coef, hipStream_t st) {
  if (sh.ld_local == 0) return;
  hipLaunchKernelGGL(gram_coef_kernel, dim3(sh.ld_local / 256), dim3(256), 0, st, RtSoA, sh.ld_local, sh.n_local, tau2, coef);
}

void launch_filter_tile(const Points& pts, const FilterTileJob& job, hipStream_t st) {
  hipLaunchKernelGGL(filter_tile_kernel, dim3((job.rows + 255) / 256), dim3(256), 0, st, pts.planes, pts.n, pts.ld, job);
}

void launch_score_filter(const Points& pts, const float* RtSoA, const float* RtAoS, const Shard& sh, const Derived& dv,
                         const FilterPlan& fp, const void* tile, void* state, void* coef, void* frame, uint32_t* partial, const Tuning& tn,
                         hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, hipEvent_t ev_mid) {
  if (sh.n_local == 0) return;
  const FilterState f = filter_state(state, fp);
  // exact pass: one thread per queue entry, grid-stride — a chain of dependent loads per entry, so big calls want more
  // threads in flight: workgroups per sub-queue by the number of tests (C4, 2.5e9 tests, 0.9 M entries: 36 -> 28 us at 8;
  // C2's 0.15 M entries are served by 2, more only adds launch tail)
  const uint64_t tests = (uint64_t)pts.n * sh.ld_local;
  const uint32_t exact_mult = tests >= (1ull << 31) ? 8u : (tests >= (1ull << 29) ? 4u : 2u);
  if (fp.mode == 2) {
    uint32_t ql = tn.filter_lds_queue ? tn.filter_lds_queue : (uint32_t)GX_QL;
    if (ql > (uint32_t)GX_QL) ql = GX_QL;
    if (ql < 64) ql = 64;
    const GramCoef gc = gram_coef_view(coef, sh.ld_local, frame);
    const uint32_t group_major = (uint64_t)((sh.ld_local + 32 * GX_WAVES - 1) / (32 * GX_WAVES)) * fp.splits <= 1024u ? 1u : 0u;  // (see the kernel)
#define SC_GRAM_LAUNCH(V)                                                                                                         \
    SC_LAUNCH_EV(score_gram_kernel<V>, dim3((sh.ld_local + 32 * GX_WAVES - 1) / (32 * GX_WAVES) * fp.splits), dim3(64 * GX_WAVES), \
                          st, ev0, ev_mid, gc, sh.ld_local, static_cast<const uint4*>(tile), fp.windows, fp.splits, fp.n_waves, partial, \
                          f.queue, f.cap_sq, f.qcount, f.redo, ql, group_major)
    // The shipped library holds variant 0 only.  -DSC_ABLATIONS (sac-cot_amd/build.py --ablations; tools/pmc_gram_variants.sh)
    // also instantiates the bit-identical scheduling variant 1 and the TIMING-ONLY bodies (no shell test / no barrier / no
    // epilogue: wrong counts by design), which sc_set_debug refuses without it.
    switch (tn.filter_variant) {
#ifdef SC_ABLATIONS
      case 512: SC_GRAM_LAUNCH(512); break;
      case 256: SC_GRAM_LAUNCH(256); break;
      case 768: SC_GRAM_LAUNCH(768); break;
      case 32: SC_GRAM_LAUNCH(32); break;
      case 288: SC_GRAM_LAUNCH(288); break;
      case 352: SC_GRAM_LAUNCH(352); break;
      case 320: SC_GRAM_LAUNCH(320); break;
      case 1: SC_GRAM_LAUNCH(1); break;
#endif
      default: SC_GRAM_LAUNCH(0); break;
    }
#undef SC_GRAM_LAUNCH
    SC_LAUNCH_EV(score_exact_kernel, dim3(FX_NQ * exact_mult), dim3(256), st, (hipEvent_t) nullptr, ev1, pts.planes, pts.n, pts.ld,
                          reinterpret_cast<const float4*>(RtAoS), sh.ld_local, dv.tau2, fp.windows, fp.splits, fp.n_waves,
                          static_cast<const uint2*>(f.queue), f.cap_sq, static_cast<const uint32_t*>(f.qcount),
                          static_cast<const uint32_t*>(f.redo), partial, static_cast<const uint32_t*>(gc.hperm),
                          static_cast<const uint32_t*>(gc.pperm), static_cast<const GramFrame*>(gc.frame));
    return;
  }
  uint32_t ql = tn.filter_lds_queue ? tn.filter_lds_queue : (uint32_t)FX_QL;
  if (ql > (uint32_t)FX_QL) ql = FX_QL;
  if (ql < 64) ql = 64;  // one step can add 64 entries
  const dim3 grid(sh.ld_local / (8 * FX_WAVES), fp.splits), block(64 * FX_WAVES);
#define SC_FILTER_LAUNCH(V)                                                                                                  \
  SC_LAUNCH_EV((score_filter_kernel<FX_WAVES, V>), grid, block, st, ev0, ev_mid, RtSoA, sh.ld_local, dv.tau2, \
                     static_cast<const uint4*>(tile), f.info, fp.windows, fp.splits, fp.n_waves, partial, f.queue, f.cap_sq, \
                     f.qcount, f.redo, ql)
  switch (tn.filter_variant) {
#ifdef SC_ABLATIONS
    case 1: SC_FILTER_LAUNCH(1); break;
    case 2: SC_FILTER_LAUNCH(2); break;
    case 3: SC_FILTER_LAUNCH(3); break;
    case 16: SC_FILTER_LAUNCH(16); break;
    case 32: SC_FILTER_LAUNCH(32); break;
    case 17: SC_FILTER_LAUNCH(17); break;
    case 33: SC_FILTER_LAUNCH(33); break;
    case 64: SC_FILTER_LAUNCH(64); break;
    case 96: SC_FILTER_LAUNCH(96); break;
    case 80: SC_FILTER_LAUNCH(80); break;
    case 128: SC_FILTER_LAUNCH(128); break;
    case 256: SC_FILTER_LAUNCH(256); break;
    case 512: SC_FILTER_LAUNCH(512); break;
    case 768: SC_FILTER_LAUNCH(768); break;
    case 288: SC_FILTER_LAUNCH(288); break;
    case 272: SC_FILTER_LAUNCH(272); break;
#endif
    default: SC_FILTER_LAUNCH(0); break;
  }
#undef SC_FILTER_LAUNCH
  SC_LAUNCH_EV(score_exact_kernel, dim3(FX_NQ * exact_mult), dim3(256), st, (hipEvent_t) nullptr, ev1, pts.planes, pts.n, pts.ld,
                        reinterpret_cast<const float4*>(RtAoS), sh.ld_local, dv.tau2, fp.windows, fp.splits, fp.n_waves,
                        static_cast<const uint2*>(f.queue), f.cap_sq, static_cast<const uint32_t*>(f.qcount),
                        static_cast<const uint32_t*>(f.redo), partial, static_cast<const uint32_t*>(nullptr),
                        static_cast<const uint32_t*>(nullptr), static_cast<const GramFrame*>(nullptr));
}

hipError_t filter_read_counters(const void* state, const FilterPlan& fp, hipStream_t st, uint64_t* undecided, uint64_t* recounts) {
  const FilterState f = filter_state(const_cast<void*>(state), fp);
  const size_t bm_words = ((size_t)fp.splits * fp.n_waves + 31) / 32;
  std::vector<uint32_t> h((size_t)FX_NQ * 32 + bm_words);
  hipError_t e = hipMemcpyAsync(h.data(), f.qcount, h.size() * 4, hipMemcpyDeviceToHost, st);  // counters, then the bitmap
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  uint64_t u = 0, r = 0;
  for (int q = 0; q < FX_NQ; q++) u += h[(size_t)q * 32];  // entries asked for (an overflowing sub-queue counts what was asked)
  for (size_t w = 0; w < bm_words; w++) r += (uint64_t)__builtin_popcount(h[(size_t)FX_NQ * 32 + w]);
  *undecided = u; *recounts = r;
  return hipSuccess;
}

hipError_t filter_read_frame(const void* frame, hipStream_t st, uint32_t out[5]) {
  std::vector<unsigned char> buf(sizeof(GramFrame));  // (8.5 KiB)
  GramFrame& f = *reinterpret_cast<GramFrame*>(buf.data());
  hipError_t e = hipMemcpyAsync(&f, frame, sizeof(GramFrame), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  uint32_t good = 0, all = 0;
  for (int v = 0; v < 64; v++) { good += (uint32_t)f.seg[v].cnt; all += (uint32_t)f.seg[v].cnt + (uint32_t)(f.seg[v].cnt >> 32); }
  out[0] = (uint32_t)f.pts; out[1] = good; out[2] = all; out[3] = f.ref_index; out[4] = f.ref_votes_q8;
  return hipSuccess;
}

// Tuning::score_split: share of the hypotheses (in 256ths) scored on the matrix pipe (default 0; experiments)
void launch_score(const Points& pts, const float* RtSoA, const float* RtAoS, const Shard& sh, const Derived& dv,
                  int score_mode, uint32_t* partial, const Tuning& tn, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  if (sh.n_local == 0) return;
  (void)RtAoS;
  uint32_t chunks;
  int chunk_pts;
  score_plan(pts.n, sh.ld_local, &chunks, &chunk_pts);
  const uint32_t groups = sh.ld_local / SCORE_THREADS;            // 256-hypothesis groups
  const uint32_t split = score_mode == 0 ? (tn.score_split <= 256 ? tn.score_split : 256u) : 0u;
  const uint32_t gm = (uint32_t)(((uint64_t)groups * split + 128) / 256);  // groups on the matrix pipe
  const uint32_t nv = groups - gm, nm = gm * (SCORE_THREADS / MF_HYPS_PER_BLOCK);
  const dim3 grid(nv + nm, chunks);
  if (score_mode == 1)
    SC_LAUNCH_EV(score_kernel<1>, grid, dim3(SCORE_THREADS), st, ev0, ev1, pts.planes, pts.n, pts.ld, RtSoA,
                          sh.ld_local, dv.inv_tau2, chunk_pts, partial, nv, nm);
  else if (score_mode == 2)
    SC_LAUNCH_EV(score_kernel<2>, grid, dim3(SCORE_THREADS), st, ev0, ev1, pts.planes, pts.n, pts.ld, RtSoA,
                          sh.ld_local, dv.inv_tau, chunk_pts, partial, nv, nm);
  else
    SC_LAUNCH_EV(score_kernel<0>, grid, dim3(SCORE_THREADS), st, ev0, ev1, pts.planes, pts.n, pts.ld, RtSoA,
                          sh.ld_local, dv.tau2, chunk_pts, partial, nv, nm);
}

uint32_t argmax_blocks(uint32_t ld_local) { return ld_local / 256 < 256 ? ld_local / 256 : 256; }
size_t argmax_scratch_bytes(uint32_t ld_local) { return (size_t)(ld_local / 256 + 1) * 2 * sizeof(uint64_t); }

void launch_argmax(const Shard& sh, const uint32_t* partial, uint32_t n_chunks, const uint32_t* sel_key,
                   uint32_t* cnt, uint64_t* pairs, uint32_t* ticket, uint64_t* key2, hipStream_t st) {
  if (sh.n_local == 0) {  // nothing scored: the pair is (0, 0)
    (void)hipMemsetAsync(key2, 0, 2 * sizeof(uint64_t), st);
    return;
  }
  const uint32_t blocks = argmax_blocks(sh.ld_local);
  hipLaunchKernelGGL(score_argmax_kernel, dim3(blocks), dim3(256), 0, st, partial, n_chunks, sh, sel_key, cnt,
                     reinterpret_cast<unsigned long long*>(pairs), ticket, reinterpret_cast<unsigned long long*>(key2));
}

// ------------------------------------------------------------------------------------------------
// C3
// ------------------------------------------------------------------------------------------------
// C3 in ONE launch: every block re-solves the winner (thread 0; deterministic, so all blocks hold the same R,t) while
// its other threads count their slice of the winner's rank index (#keys above the winner's + #equal keys at lower
// positions — the number the ranked list would have given it), then masks its 256 correspondences.  The slice counts
// meet in a control-block counter; the block that takes the last ticket publishes (key, position, rank) to the host.
// (A single-block rank count cost 42 us at T = 400 k; a separate mask launch another ~4.5 us floor.)
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ planes, int n, int ld,
                                                       TriSource ts, Shard sh, const float* __restrict__ RtSoA,
                                                       const uint32_t* __restrict__ sel_key, uint32_t T,
                                                       const unsigned long long* __restrict__ key2, int npairs,
                                                       unsigned long long* __restrict__ key_out, float tau2,
                                                       float* __restrict__ Rt12, uint8_t* __restrict__ mask,
                                                       unsigned long long* __restrict__ fin_word,
                                                       unsigned long long* __restrict__ host_out, DeferredPub dp) {
  __shared__ uint64_t lds[8];
  __shared__ float sRt[12];
  __shared__ uint32_t s_last;
  // (what does not depend on the winner is on its way before the pairs are looked at: this thread's correspondence, its first
  // keys of the rank count — the kernel is a chain of dependent loads, ~1 us each)
  const int m = blockIdx.x * 256 + threadIdx.x;
  float cp[6];
#pragma unroll
  for (int c = 0; c < 6; c++) cp[c] = m < n ? planes[(size_t)c * ld + m] : 0.f;
  const uint32_t T4 = T >> 2;  // 16-byte loads, grid-strided
  const uint4* __restrict__ k4 = reinterpret_cast<const uint4*>(sel_key);
  const uint32_t q_first = blockIdx.x * 256 + threadIdx.x;
  uint4 v_first = make_uint4(0u, 0u, 0u, 0u);
  if (sel_key != nullptr && q_first < T4) v_first = k4[q_first];
  // key2: npairs winner key pairs (one per rank, all-gathered; npairs = 1: an already reduced pair).  The reduction of
  // include/saccot.h — K0 = max pair[0], K1 = max pair[1] among the pairs attaining K0 — is a lexicographic max.
  unsigned long long k0 = 0, k1 = 0;
  if (npairs <= 8) {
    for (int w = 0; w < npairs; w++) {  // wave-uniform addresses: scalar loads
      const unsigned long long a = key2[2 * w], b = key2[2 * w + 1];
      if (a > k0 || (a == k0 && b > k1)) { k0 = a; k1 = b; }
    }
  } else {  // many pairs (the arg-max launch's own workgroups', or a large world's): a thread takes every 256th, the block reduces
    for (int w = threadIdx.x; w < npairs; w += 256) {
      const unsigned long long a = key2[2 * w], b = key2[2 * w + 1];
      if (a > k0 || (a == k0 && b > k1)) { k0 = a; k1 = b; }
    }
    unsigned long long* l4 = reinterpret_cast<unsigned long long*>(lds);
    const unsigned long long K = block_max_u64(k0, l4);
    __syncthreads();
    const unsigned long long P = block_max_u64(k0 == K ? k1 : 0ull, l4);
    __syncthreads();
    k0 = K; k1 = P;
  }
  const bool two_stage = sel_key != nullptr;
  uint32_t g = 0;
  if (k0 != 0) g = 0xFFFFFFFFu - (uint32_t)((two_stage ? k1 : k0) & 0xFFFFFFFFull);
  // The pairs come from the caller (an all-gather): a position outside the selection — a stale or uninitialised pair,
  // ranks that disagree on T or the parameters — must not index sel_key / the triangle lookup.  It is treated as "no
  // hypothesis" (identity, zero mask) and reported: host_out[1] = all ones makes the host return SC_EINVAL.
  const bool bad_pair = k0 != 0 && g >= T;
  if (bad_pair) { k0 = 0; k1 = 0; g = 0; }
  if (key_out && blockIdx.x == 0 && threadIdx.x == 0) { key_out[0] = k0; key_out[1] = k0 ? k1 : 0ull; }
  // The winner's (R,t): if THIS rank scored it, phase 1 left it in RtSoA (kabsch3 is deterministic, so these are the
  // very bits a re-solve gives) — 12 parallel loads; otherwise thread 0 re-solves it from the replicated selection.
  const uint32_t gb = sh.block ? g / sh.block : 0u;
  const bool local = RtSoA != nullptr && k0 != 0 && g < sh.T_eff && (gb % sh.world) == sh.rank;
  if (local) {
    const uint32_t l = (gb / sh.world) * sh.block + (g % sh.block);
    if (threadIdx.x < 12) {
      const float v = RtSoA[(size_t)threadIdx.x * sh.ld_local + l];
      sRt[threadIdx.x] = v;
      if (blockIdx.x == 0) Rt12[threadIdx.x] = v;
    }
  } else if (threadIdx.x == 0) {
    float Rt[12] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};
    uint32_t v[3];
    if (k0 != 0 && tri_lookup(ts, g, v)) {
      float P[9], Q[9];
      load_triangle(planes, ld, v, P, Q);
      kabsch3(P, Q, Rt);
    }
#pragma unroll
    for (int c = 0; c < 12; c++) sRt[c] = Rt[c];
    if (blockIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < 12; c++) Rt12[c] = Rt[c];
    }
  }
  uint32_t r = 0;
  if (two_stage && k0 != 0) {
    const uint32_t wk = (uint32_t)(k0 & 0xFFFFFFFFull);  // = sel_key[g]: the key's low half (score_argmax_kernel)
    auto rank4 = [&](const uint4& v, uint32_t q) {
      const uint32_t t = q << 2;
      r += (v.x > wk) || (v.x == wk && t < g);
      r += (v.y > wk) || (v.y == wk && t + 1 < g);
      r += (v.z > wk) || (v.z == wk && t + 2 < g);
      r += (v.w > wk) || (v.w == wk && t + 3 < g);
    };
    const uint32_t qs = gridDim.x * 256;
    if (q_first < T4) rank4(v_first, q_first);
    uint32_t q = q_first + qs;
    for (; q + 3 * qs < T4; q += 4 * qs) {  // (four loads in flight: see score_argmax_kernel)
      const uint4 a0 = k4[q], a1 = k4[q + qs], a2 = k4[q + 2 * qs], a3 = k4[q + 3 * qs];
      rank4(a0, q); rank4(a1, q + qs); rank4(a2, q + 2 * qs); rank4(a3, q + 3 * qs);
    }
    {
      uint4 w[3];
#pragma unroll
      for (uint32_t u = 0; u < 3; u++) w[u] = q + u * qs < T4 ? k4[q + u * qs] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (uint32_t u = 0; u < 3; u++) if (q + u * qs < T4) rank4(w[u], q + u * qs);
    }
    if (blockIdx.x == 0) {
      const uint32_t t = (T4 << 2) + threadIdx.x;  // the last T % 4 keys
      if (t < T) { const uint32_t kt = sel_key[t]; r += (kt > wk) || (kt == wk && t < g); }
    }
  }
  const uint64_t rb = block_reduce_u64(r, lds);  // also the barrier that publishes sRt to the block
  if (m < n) {
    float M[12];
#pragma unroll
    for (int c = 0; c < 12; c++) M[c] = sRt[c];
    const bool live = k0 != 0ull && finite12(M);
    const float d2 = resid2(M, cp[0], cp[1], cp[2], cp[3], cp[4], cp[5]);
    mask[m] = (live && d2 < tau2) ? 1 : 0;
  }
  if (threadIdx.x == 0) {
    // ONE returning atomic per workgroup: workgroups finished in the high half, the rank count so far in the low half — the workgroup
    // that finds every other one finished holds the whole count in the value it got back (no second accumulator, no acquire, no load)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned long long was = __hip_atomic_fetch_add(fin_word, (1ull << 32) | (unsigned long long)(uint32_t)rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ((uint32_t)(was >> 32) == gridDim.x - 1) ? 1u : 0u;
    if (s_last) {
      const uint32_t rank = (uint32_t)was + (uint32_t)rb;
      __hip_atomic_store(fin_word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
      if (host_out && dp.host) {
        // a host-free call: the words its earlier kernels would have published one by one go to the host HERE, with the winner
        // (relaxed system-scope stores: the release store of the key below orders them before it).  Every such store costs the
        // kernel that makes it ~0.5 us (seven of them cost the staging kernel 1.7 us; nine here cost this kernel 5:
        // profiles/r05_ab_deferred_publish.txt), so stage B's two counts travel in ONE word — edges in the low half, triangles in
        // the high half, all ones where one of them does not fit (the host then repeats the call) — and the staging kernel's
        // coordinate statistics only every 64th host-free call of a context (dp.with_stats): a host-free call picks stage C2's
        // kernel by the statistics of an earlier frame anyway, and any pick gives the same counts.
        auto put = [&](int idx, unsigned long long v) { __hip_atomic_store(&dp.host[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
        const unsigned long long e = *dp.dev_edges, m = *dp.dev_triangles;
        put(0, (e < (1ull << 32) && m < (1ull << 32)) ? (e | (m << 32)) : ~0ull);
        if (dp.with_stats) {
          for (int k = 0; k < 6; k++) put(16 + k, ((unsigned long long)dp.coord_max[8 + k] << 32) | dp.coord_max[2 + k]);
          put(13, ((unsigned long long)dp.coord_max[1] << 32) | dp.coord_max[0]);
        }
      }
      if (host_out) {  // [0] last: the host polls it (release orders the others before it)
        // ONE word beside the key (a system-scope store costs ~0.5 us: see DeferredPub): the winner's rank index in the high half,
        // its position in the low half — or all ones: a key pair that decodes to nothing of the selection (the host: SC_EINVAL)
        const unsigned long long rk = k0 ? (two_stage ? (unsigned long long)rank : (unsigned long long)g) : 0ull;
        __hip_atomic_store(&host_out[1], bad_pair ? ~0ull : ((rk << 32) | (unsigned long long)g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        publish_host(reinterpret_cast<uint64_t*>(host_out), k0);
      }
    }
  }
}

__global__ __launch_bounds__(256) void mask_kernel(const float* __restrict__ planes, int n, int ld,
                                                   const float* __restrict__ Rt12,
                                                   const unsigned long long* __restrict__ key, float tau2,
                                                   uint8_t* __restrict__ mask) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  float M[12];
#pragma unroll
  for (int c = 0; c < 12; c++) M[c] = Rt12[c];
  const bool live = (key == nullptr || *key != 0ull) && finite12(M);
  const float d2 = resid2(M, planes[m], planes[(size_t)ld + m], planes[2 * (size_t)ld + m],
                          planes[3 * (size_t)ld + m], planes[4 * (size_t)ld + m], planes[5 * (size_t)ld + m]);
  mask[m] = (live && d2 < tau2) ? 1 : 0;
}

void launch_finalize(const Points& pts, const TriSource& ts, const Shard& sh, const float* RtSoA,
                     const uint32_t* sel_key, uint32_t T, const uint64_t* key2, int npairs, uint64_t* key_out, float tau2, float* Rt12, uint8_t* mask,
                     unsigned long long* fin_word, uint64_t* host_out, hipStream_t st, const DeferredPub* dp) {
  uint32_t blocks = (uint32_t)((pts.n + 255) / 256);  // the mask needs these; more only if the key list is long
  const uint32_t for_keys = (T / 4 + 1023) / 1024;    // >= 4 uint4 per thread before another block pays off
  if (for_keys > blocks) blocks = for_keys < 1024u ? for_keys : 1024u;
  hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, st, pts.planes, pts.n, pts.ld, ts, sh, RtSoA, sel_key, T,
                     reinterpret_cast<const unsigned long long*>(key2), npairs,
                     reinterpret_cast<unsigned long long*>(key_out), tau2, Rt12, mask, fin_word,
                     reinterpret_cast<unsigned long long*>(host_out), dp ? *dp : DeferredPub{nullptr, nullptr, nullptr, nullptr, 0});
}

// ------------------------------------------------------------------------------------------------
// Winner refinement (optional — SC_FLAG_REFINE): fp64 least-squares rigid refit over the inlier mask.
// Canonical order shared with oracle/saccot_oracle.c::so_refine: chunks of 64 consecutive points are summed
// sequentially in index order (one thread per chunk), chunk sums are added sequentially in chunk order (thread 0);
// pass 1 gives count and centroids, pass 2 H = sum (p - pc)(q - qc)^T by fma, then the two-dominant-pairs +
// cross-product construction of kabsch3 in double with 10 Jacobi sweeps.  One workgroup: N is a few thousand.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ddot3(const double* a, const double* b) {
  return __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], a[0] * b[0]));
}
__device__ __forceinline__ void dcross3(const double* a, const double* b, double* c) {
  c[0] = __builtin_fma(a[1], b[2], -(a[2] * b[1]));
  c[1] = __builtin_fma(a[2], b[0], -(a[0] * b[2]));
  c[2] = __builtin_fma(a[0], b[1], -(a[1] * b[0]));
}

__global__ __launch_bounds__(1024) void refine_kernel(const float* __restrict__ planes, int n, int ld,
                                                      const uint8_t* __restrict__ mask,
                                                      const unsigned long long* __restrict__ key2,
                                                      double* __restrict__ scratch, float* __restrict__ Rt12) {
  __shared__ double cen[8];
  if (key2[0] == 0ull) return;  // no hypothesis: nothing to refine (uniform)
  const int nch = (n + 63) / 64;
  // pass 1: per-chunk count / sum p / sum q
  for (int ch = threadIdx.x; ch < nch; ch += 1024) {
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    const int m1 = min(n, ch * 64 + 64);
    for (int m = ch * 64; m < m1; m++) {
      if (!mask[m]) continue;
      c[0] += 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) { c[1 + k] += (double)planes[(size_t)k * ld + m]; c[4 + k] += (double)planes[(size_t)(3 + k) * ld + m]; }
    }
#pragma unroll
    for (int k = 0; k < 7; k++) scratch[(size_t)ch * 16 + k] = c[k];
  }
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) {
    double S[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < nch; ch++)
#pragma unroll
      for (int k = 0; k < 7; k++) S[k] += scratch[(size_t)ch * 16 + k];
    cen[0] = S[0];
#pragma unroll
    for (int k = 0; k < 3; k++) { cen[1 + k] = S[1 + k] / S[0]; cen[4 + k] = S[4 + k] / S[0]; }
  }
  __syncthreads();
  if (cen[0] < 3.0) return;  // uniform
  const double pc[3] = {cen[1], cen[2], cen[3]}, qc[3] = {cen[4], cen[5], cen[6]};
  // pass 2: per-chunk covariance
  for (int ch = threadIdx.x; ch < nch; ch += 1024) {
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int m1 = min(n, ch * 64 + 64);
    for (int m = ch * 64; m < m1; m++) {
      if (!mask[m]) continue;
      double a[3], b[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        a[k] = (double)planes[(size_t)k * ld + m] - pc[k];
        b[k] = (double)planes[(size_t)(3 + k) * ld + m] - qc[k];
      }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) h[3 * r + c] = __builtin_fma(a[r], b[c], h[3 * r + c]);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) scratch[(size_t)ch * 16 + k] = h[k];
  }
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x != 0) return;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int ch = 0; ch < nch; ch++)
#pragma unroll
    for (int k = 0; k < 9; k++) H[k] += scratch[(size_t)ch * 16 + k];
  double B[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) B[c][r] = H[3 * r + c];
#pragma unroll 1
  for (int sweep = 0; sweep < 10; sweep++) {
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
      const int ip = (pr == 2) ? 1 : 0, iq = (pr == 0) ? 1 : 2;
      const double alpha = ddot3(B[ip], B[ip]), beta = ddot3(B[iq], B[iq]), gamma = ddot3(B[ip], B[iq]);
      if (gamma != 0.0) {
        const double zeta = (beta - alpha) / (gamma + gamma);
        double tt = 1.0 / (__builtin_fabs(zeta) + __builtin_sqrt(__builtin_fma(zeta, zeta, 1.0)));
        if (zeta < 0.0) tt = -tt;
        const double cs = 1.0 / __builtin_sqrt(__builtin_fma(tt, tt, 1.0));
        const double sn = cs * tt;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double x = B[ip][r], y = B[iq][r];
          B[ip][r] = __builtin_fma(-sn, y, cs * x);
          B[iq][r] = __builtin_fma(sn, x, cs * y);
          x = V[ip][r]; y = V[iq][r];
          V[ip][r] = __builtin_fma(-sn, y, cs * x);
          V[iq][r] = __builtin_fma(sn, x, cs * y);
        }
      }
    }
  }
  const double n0 = ddot3(B[0], B[0]), n1 = ddot3(B[1], B[1]), n2 = ddot3(B[2], B[2]);
  int i1 = 0; double m1 = n0;
  if (n1 > m1) { i1 = 1; m1 = n1; }
  if (n2 > m1) { i1 = 2; m1 = n2; }
  int i2 = (i1 == 0) ? 1 : 0;
  {
    const int c = 3 - i1 - i2;
    const double nc = (c == 0) ? n0 : (c == 1 ? n1 : n2), ni2 = (i2 == 0) ? n0 : (i2 == 1 ? n1 : n2);
    if (nc > ni2) i2 = c;
  }
  double b1[3], b2[3], v1[3], v2[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    b1[r] = (i1 == 0) ? B[0][r] : (i1 == 1 ? B[1][r] : B[2][r]);
    b2[r] = (i2 == 0) ? B[0][r] : (i2 == 1 ? B[1][r] : B[2][r]);
    v1[r] = (i1 == 0) ? V[0][r] : (i1 == 1 ? V[1][r] : V[2][r]);
    v2[r] = (i2 == 0) ? V[0][r] : (i2 == 1 ? V[1][r] : V[2][r]);
  }
  const double s1 = __builtin_sqrt(ddot3(b1, b1)), s2 = __builtin_sqrt(ddot3(b2, b2));
  double u1[3], u2[3], u3[3], v3[3];
#pragma unroll
  for (int r = 0; r < 3; r++) { u1[r] = b1[r] / s1; u2[r] = b2[r] / s2; }
  dcross3(u1, u2, u3);
  dcross3(v1, v2, v3);
  double R[9], t[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) R[3 * r + c] = __builtin_fma(v3[r], u3[c], __builtin_fma(v2[r], u2[c], v1[r] * u1[c]));
#pragma unroll
  for (int r = 0; r < 3; r++)
    t[r] = qc[r] - __builtin_fma(R[3 * r + 2], pc[2], __builtin_fma(R[3 * r + 1], pc[1], R[3 * r] * pc[0]));
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 9; k++) ok = ok && (__builtin_fabs(R[k]) < __builtin_inf());
#pragma unroll
  for (int k = 0; k < 3; k++) ok = ok && (__builtin_fabs(t[k]) < __builtin_inf());
  if (!ok) return;
#pragma unroll
  for (int k = 0; k < 9; k++) Rt12[k] = (float)R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) Rt12[9 + k] = (float)t[k];
}

size_t refine_scratch_bytes(int n) { return (size_t)((n + 63) / 64) * 16 * sizeof(double); }

void launch_refine(const Points& pts, const uint8_t* mask, const uint64_t* key2, double* scratch, float* Rt12,
                   hipStream_t st) {
  hipLaunchKernelGGL(refine_kernel, dim3(1), dim3(1024), 0, st, pts.planes, pts.n, pts.ld, mask,
                     reinterpret_cast<const unsigned long long*>(key2), scratch, Rt12);
}

void launch_mask(const Points& pts, const float* Rt12, float tau2, uint8_t* mask, hipStream_t st) {
  hipLaunchKernelGGL(mask_kernel, dim3((pts.n + 255) / 256), dim3(256), 0, st, pts.planes, pts.n, pts.ld, Rt12,
                     (const unsigned long long*)nullptr, tau2, mask);
}

}  // namespace sc
