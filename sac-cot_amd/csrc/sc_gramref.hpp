// sc_gramref.hpp — the reference frame of the Gram filter (stage C2): its layout and the workgroup that votes for it.
// Shared by sc_score.hip (which makes tile and coefficients in this frame, and votes in a launch of its own where nothing else
// can) and sc_tri.hip (whose counting pass carries the vote as an extra workgroup on the hot path).  Device code only.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include "sc_arith.hpp"
#include "sc_kernels.hpp"

namespace sc {

__host__ __device__ __forceinline__ uint32_t shard_global_index(uint32_t l, uint32_t block, uint32_t rank,
                                                                uint32_t world) {
  return ((l / block) * world + rank) * block + (l % block);
}

constexpr double GX_KAPPA = 8.0;     // a hypothesis is NEAR the reference while its reach (below) stays under GX_KAPPA tau'
constexpr int GX_VOTE = 64;          // hypotheses that vote for the reference frame (gram_ref_block)
// The frame of one call's Gram filter: made by gram_ref_block (one workgroup: a launch of its own before the Kabsch launch, or an
// extra workgroup of stage B's counting pass), read by every workgroup of
// the Kabsch launch (tile and coefficients), by the filter and by the exact pass.  A buffer of its own per context (128-aligned).
struct GramFrame {
  double R0[9], t0[3];  // the reference motion: R0 orthogonal to ~1e-16 (rebuilt in fp64 from a unit quaternion), t0
  double c[3];          // the common centre: centre of the source cloud's bounding box
  double s;             // power of two: every |coordinate| of P' = s (p - c) and of Q' = s (R0^T (q - t0) - c) stays below 64
  double Pn, Qn;        // upper bounds of |P'| and |Q'| (Euclidean); |V'| = |Q' - P'| <= Vn = Pn + Qn
  double st;            // s sqrt(tau2)
  double th_h;          // a hypothesis is NEAR the reference when reach_h + dE_h <= th_h  (= GX_KAPPA st)
  double th_v;          // a correspondence is NEAR when |V'| <= th_v = th_h + 1.05 st + 1e-3: a FAR one cannot be an inlier of a NEAR hypothesis
  float pmax_o, qmax_o; // max |coordinate| of the original clouds: what the canonical chain rounds at
  uint32_t ref_votes_q8, ref_index;       // diagnostics: the winner's (soft) vote count x 256, its position in this rank's shard
  uint32_t pad0[2];
  // Class counters, filled by the Kabsch launch's atomics; each on a 128-byte line of its own, away from what every workgroup of that
  // launch reads (the frame is 128-aligned).  A counter packs two counts: low word from the front, high word from the back.
  //   pts      low: NEAR correspondences (tile rows [0, near)), high: FAR ones (rows n - 1 downwards)
  //   seg[s]   the coefficient rows are dealt in GX_SEGS(groups) interleaved segments — workgroup b of 256 hypotheses belongs to
  //            segment b % S, whose q-th block of 256 rows is row block q S + s — so that the S chains of dependent same-address
  //            atomics are S times shorter (one chain cost ~45 ns per workgroup: 180 us at C4's 1954 workgroups).  low: hypotheses
  //            NEAR the reference (the segment's rows from its front), high: the others (from its back).
  alignas(128) unsigned long long pts; uint32_t pad1[30];
  struct Seg { unsigned long long cnt; uint32_t pad[30]; } seg[64];
};
static_assert(offsetof(GramFrame, pts) == 256 && offsetof(GramFrame, seg) == 384 && sizeof(GramFrame) == 384 + 64 * 128, "frame layout");
__host__ __device__ inline uint32_t gram_segments(uint32_t groups) {  // S: ~32 workgroups per chain, at most 64 chains
  const uint32_t s = groups / 32u;
  return s < 1u ? 1u : (s > 64u ? 64u : s);
}
// row of the v-th row (from the front) of segment s
__host__ __device__ inline uint32_t gram_seg_row(uint32_t v, uint32_t s, uint32_t S) { return ((v >> 8) * S + s) * 256u + (v & 255u); }
// what a workgroup keeps of the frame (LDS copy: one read of global memory per workgroup)
struct GramFrameRO { double R0[9], t0[3], c[3], s, Pn, Qn, st, th_h, th_v; float pmax_o, qmax_o; };

// the three correspondences of a triangle, from the AoS copy behind the planes (8 floats each: two 16-byte loads per
// vertex instead of six scattered 4-byte gathers)
__device__ __forceinline__ void load_triangle(const float* __restrict__ planes, int ld, const uint32_t* tri3,
                                              float P[9], float Q[9]) {
  const float4* __restrict__ aos4 = reinterpret_cast<const float4*>(planes + 6 * (size_t)ld);
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const uint32_t v = tri3[m];
    const float4 a = aos4[2 * (size_t)v], b = aos4[2 * (size_t)v + 1];
    P[3 * m] = a.x; P[3 * m + 1] = a.y; P[3 * m + 2] = a.z;
    Q[3 * m] = a.w; Q[3 * m + 1] = b.x; Q[3 * m + 2] = b.y;
  }
}

// triangle g of the selected list, straight from the selection (two dependent lookups, no materialised list)
// false: the selection holds something that must not be followed (TriSource::lim_*: host-free calls that will be repeated)
__device__ __forceinline__ bool tri_lookup(const TriSource& ts, uint32_t g, uint32_t v[3]) {
  if (ts.cand_recs) {  // sharded stage B: the record travels with the candidate
    const uint64_t pos = ts.sel_ord[g], sg = pos / ts.cand_seg;
    const uint4 rec = ts.cand_recs[sg * ts.cand_stride + (pos - sg * ts.cand_seg)];
    v[0] = rec.x; v[1] = rec.y; v[2] = rec.z;
    return true;
  }
  const uint64_t ord = ts.sel_ord[g];
  if (ts.lim_ord && ord >= ts.lim_ord) return false;
  const uint2 ke = ts.kcol[ord];
  if (ts.lim_edge && ke.y >= ts.lim_edge) return false;
  v[0] = ts.ei[ke.y];
  v[1] = ts.ej[ke.y];
  v[2] = ke.x;
  return !ts.lim_vertex || (v[0] < ts.lim_vertex && v[1] < ts.lim_vertex && v[2] < ts.lim_vertex);
}


// Where the 64 voters come from — one of:
//   RtSoA   hypotheses that exist already (stage hook sc_score_host): voter k = hypothesis floor(k n_local / 64)
//   ts      the selection (after stage B): voter k = the hypothesis of ranked triangle floor(k n_local / 64) of this rank's shard
//   cand    candidate triangles {key bits, i, j, k} found by the estimating sample of stage B, one per workgroup of that launch
//           (0 key: none): voter k = the best-keyed candidate of the workgroups = k (mod 64) (cand_slot[k] names it).  This is what lets the
//           vote run long BEFORE the selection exists, as an extra workgroup of the counting pass (sc_tri.hip): the best of a few
//           hundred sampled triangles sits around the top-T boundary, and that is all a voter needs to be.
struct GramRefSrc {
  const float* RtSoA;
  TriSource ts; Shard sh; const uint64_t* t_eff_dev;
  const uint4* cand; uint32_t n_cand; const unsigned long long* cand_slot;
};
struct GramRefJob {  // (all null / 0: no vote)
  const float* planes; int n, ld;
  GramRefSrc src;
  const uint32_t* mx; float tau2, kappa;
  GramFrame* out;
};

// The reference frame of this call: ONE workgroup of 256 threads (every thread must call).  The 64 lanes of wave 0 each bring one
// voter; the four waves compare every voter with 16 of the others each, through LDS; the voter with the largest soft count of
// agreeing voters (its displacement field differs from the other's by less than ~6 tau anywhere in the source cloud's box; ties:
// the lower lane) becomes (R0, t0).  Thread 0 then writes the frame and clears the class counters.  No finite voter, or
// tau = 0: the frame that moves the centre of one box onto the other's, which is r03's form.
constexpr int GX_REF_LDS_WORDS = GX_VOTE * 12 + 4 * GX_VOTE;  // LDS the vote needs (4 KiB): the caller lends it
__device__ inline void gram_ref_block(const GramRefJob& job, float* __restrict__ lds) {
  static_assert(GX_VOTE == 64, "one wave of voters");
  float (*sR)[12] = reinterpret_cast<float (*)[12]>(lds);
  float (*s_score)[GX_VOTE] = reinterpret_cast<float (*)[GX_VOTE]>(lds + GX_VOTE * 12);
  const float* __restrict__ planes = job.planes;
  const int n = job.n, ld = job.ld;
  const uint32_t* __restrict__ mx = job.mx;
  const float tau2 = job.tau2, kappa = job.kappa;
  GramFrame* __restrict__ out = job.out;
  const Shard& sh = job.src.sh;
  const uint32_t k = threadIdx.x & 63u, part = threadIdx.x >> 6;  // wave `part` compares every voter with 16 of the others
  float Rt[12];
  bool fin = false;
  uint32_t my_index = 0xFFFFFFFFu;
  if (part == 0) {
    bool have = false;
    if (job.src.RtSoA) {
      const uint32_t l = (uint32_t)(((uint64_t)k * sh.n_local) / GX_VOTE);  // (n_local > 0: the host does not ask otherwise)
#pragma unroll
      for (int c = 0; c < 12; c++) Rt[c] = job.src.RtSoA[(size_t)c * sh.ld_local + l];
      have = true; my_index = l;
    } else {
      uint32_t v[3] = {0u, 0u, 0u};
      bool tri = false;
      if (job.src.cand) {
        const unsigned long long sl = job.src.cand_slot[k];  // best (key << 32 | workgroup) among the sample's workgroups = k (mod 64); 0: none
        const uint32_t c = (uint32_t)sl;
        if (sl != 0ull && c < job.src.n_cand) {
          const uint4 e = job.src.cand[c];
          v[0] = e.y; v[1] = e.z; v[2] = e.w; my_index = c;
          // (never follow an index beyond this call's n, whatever the candidate list holds)
          tri = e.x != 0u && v[0] < (uint32_t)n && v[1] < (uint32_t)n && v[2] < (uint32_t)n;
        }
      } else {
        const uint32_t l = (uint32_t)(((uint64_t)k * sh.n_local) / GX_VOTE);
        const uint32_t g = shard_global_index(l, sh.block, sh.rank, sh.world);
        tri = (!job.src.t_eff_dev || (uint64_t)g < *job.src.t_eff_dev) && tri_lookup(job.src.ts, g, v);
        my_index = l;
      }
      if (tri) {
        float P[9], Q[9];
        load_triangle(planes, ld, v, P, Q);
        kabsch3(P, Q, Rt);
        have = true;
      }
    }
    if (!have) {
#pragma unroll
      for (int c = 0; c < 12; c++) Rt[c] = 0.0f;
    }
    fin = have && finite12(Rt);
#pragma unroll
    for (int c = 0; c < 12; c++) sR[k][c] = Rt[c];
  }
  // the boxes (keys as the staging kernel leaves them: [2 + c] max, [8 + c] -min)
  double cP[3], cQ[3], hP[3], hQ[3];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const float hi = float_unkey(mx[2 + c]), lo = -float_unkey(mx[8 + c]);
    const float ctr = 0.5f * hi + 0.5f * lo;
    const double a = (double)hi - (double)ctr, b = (double)ctr - (double)lo;
    (c < 3 ? cP[c] : cQ[c - 3]) = (double)ctr;
    (c < 3 ? hP[c] : hQ[c - 3]) = a > b ? a : b;
  }
  const float Pr2 = (float)(hP[0] * hP[0] + hP[1] * hP[1] + hP[2] * hP[2]);
  const float cx = (float)cP[0], cy = (float)cP[1], cz = (float)cP[2];
  __syncthreads();
  if (part != 0) {
#pragma unroll
    for (int c = 0; c < 12; c++) Rt[c] = sR[k][c];
  }
  // agreement of voters i and j: x = |R_i - R_j|_F^2 |half diagonal|^2 + |(R_i - R_j) c + t_i - t_j|^2 (the squared displacement
  // difference anywhere in the source box is at most twice that) against (6 tau)^2; weight 1 - x / (6 tau)^2: no root, no division
  const float inv = tau2 > 0.f ? 1.0f / (36.0f * tau2) : 0.0f;
  float score = 0.f;
  for (int j = 16 * (int)part; j < 16 * (int)part + 16; j++) {
    float d2 = 0.f, dv2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float a0 = Rt[3 * i] - sR[j][3 * i], a1 = Rt[3 * i + 1] - sR[j][3 * i + 1], a2 = Rt[3 * i + 2] - sR[j][3 * i + 2];
      d2 += a0 * a0 + a1 * a1 + a2 * a2;
      const float dv = a0 * cx + a1 * cy + a2 * cz + (Rt[9 + i] - sR[j][9 + i]);
      dv2 += dv * dv;
    }
    const float w = 1.0f - (d2 * Pr2 + dv2) * inv;
    score += w > 0.f ? w : 0.f;  // (NaN: no vote)
  }
  s_score[part][k] = score;
  __syncthreads();
  if (part != 0) return;
  score = (s_score[0][k] + s_score[1][k]) + (s_score[2][k] + s_score[3][k]);
  if (!fin) score = -1.0f;
  float bs = score;
  uint32_t bl = k;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float os = __shfl_xor(bs, o);
    const uint32_t ol = (uint32_t)__shfl_xor((int)bl, o);
    if (os > bs || (os == bs && ol < bl)) { bs = os; bl = ol; }
  }
  const uint32_t win_index = (uint32_t)__shfl((int)my_index, (int)bl);
  out->seg[k].cnt = 0ull;  // (wave 0, one lane per segment)
  if (k != 0) return;
  out->pts = 0ull;
  double q[4] = {1.0, 0.0, 0.0, 0.0}, t0[3];
  bool ok = bs > 0.f;
  if (ok) {
    const float* W = sR[bl];
    const double m00 = W[0], m01 = W[1], m02 = W[2], m10 = W[3], m11 = W[4], m12 = W[5], m20 = W[6], m21 = W[7], m22 = W[8];
    const double tr = m00 + m11 + m22;
    double S;
    if (tr > 0.0) { S = sqrt(tr + 1.0) * 2.0; q[0] = 0.25 * S; q[1] = (m21 - m12) / S; q[2] = (m02 - m20) / S; q[3] = (m10 - m01) / S; }
    else if (m00 > m11 && m00 > m22) { S = sqrt(1.0 + m00 - m11 - m22) * 2.0; q[0] = (m21 - m12) / S; q[1] = 0.25 * S; q[2] = (m01 + m10) / S; q[3] = (m02 + m20) / S; }
    else if (m11 > m22) { S = sqrt(1.0 + m11 - m00 - m22) * 2.0; q[0] = (m02 - m20) / S; q[1] = (m01 + m10) / S; q[2] = 0.25 * S; q[3] = (m12 + m21) / S; }
    else { S = sqrt(1.0 + m22 - m00 - m11) * 2.0; q[0] = (m10 - m01) / S; q[1] = (m02 + m20) / S; q[2] = (m12 + m21) / S; q[3] = 0.25 * S; }
    const double nn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (nn > 1e-12 && nn < 1e12) {
      const double inv = 1.0 / sqrt(nn);
      q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
      t0[0] = W[9]; t0[1] = W[10]; t0[2] = W[11];
    } else {
      ok = false;
    }
  }
  if (!ok) {
    q[0] = 1.0; q[1] = q[2] = q[3] = 0.0;
    for (int c = 0; c < 3; c++) t0[c] = cQ[c] - cP[c];
  }
  GramFrame& f = *out;  // (written in place: the struct carries 8 KiB of counter lines)
  {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    f.R0[0] = 1.0 - 2.0 * (y * y + z * z); f.R0[1] = 2.0 * (x * y - z * w); f.R0[2] = 2.0 * (x * z + y * w);
    f.R0[3] = 2.0 * (x * y + z * w); f.R0[4] = 1.0 - 2.0 * (x * x + z * z); f.R0[5] = 2.0 * (y * z - x * w);
    f.R0[6] = 2.0 * (x * z - y * w); f.R0[7] = 2.0 * (y * z + x * w); f.R0[8] = 1.0 - 2.0 * (x * x + y * y);
  }
  double off2 = 0.0;
  for (int i = 0; i < 3; i++) {
    f.t0[i] = t0[i];
    f.c[i] = cP[i];
    const double o = cQ[i] - t0[i] - (f.R0[3 * i] * cP[0] + f.R0[3 * i + 1] * cP[1] + f.R0[3 * i + 2] * cP[2]);
    off2 += o * o;
  }
  const double hPn = sqrt(hP[0] * hP[0] + hP[1] * hP[1] + hP[2] * hP[2]);
  const double Qno = sqrt(hQ[0] * hQ[0] + hQ[1] * hQ[1] + hQ[2] * hQ[2]) + sqrt(off2);  // |R0^T (q - t0) - c| <= this
  double hmax = hP[0] > hP[1] ? hP[0] : hP[1];
  hmax = hP[2] > hmax ? hP[2] : hmax;
  hmax = Qno > hmax ? Qno : hmax;
  int e = 0;
  if (hmax > 0.0) {  // hmax in [2^e, 2^(e+1))   (inf / NaN: e = 1024, clamped below)
    union { double d; uint64_t u; } xx; xx.d = hmax;
    e = (int)((xx.u >> 52) & 2047u) - 1023;
  }
  int kk = 5 - e;  // s hmax in [32, 64)
  kk = kk > 100 ? 100 : (kk < -100 ? -100 : kk);
  union { double d; uint64_t u; } sc; sc.u = (uint64_t)(kk + 1023) << 52;
  f.s = sc.d;
  f.Pn = f.s * hPn * (1.0 + 2e-6);
  f.Qn = f.s * Qno * (1.0 + 2e-6);
  f.st = f.s * (double)sqrt_rn(tau2);
  f.th_h = (double)kappa * f.st;
  f.th_v = f.th_h + 1.05 * f.st + 1e-3;
  f.pmax_o = __uint_as_float(mx[0]); f.qmax_o = __uint_as_float(mx[1]);
  f.ref_votes_q8 = ok ? (uint32_t)(bs * 256.0f) : 0u;
  f.ref_index = ok ? win_index : 0xFFFFFFFFu;
}


}  // namespace sc
