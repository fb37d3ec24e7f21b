// sc_tri.hip — stage B: triangles_topT on the GPU.
//
// Total order of the output: key descending, then (i,j,k) lexicographic ascending.  The trick that makes the
// tie-break free: triangles are enumerated edge by edge in CSR order (i asc, j asc) and, inside an edge, in
// ascending k, so a triangle's position in that enumeration — its *ordinal* — orders like (i,j,k).  So:
//   1. edge_fill       CSR list of upper-triangle edges (ei, ej, es = s_ij) from the bit rows
//   2. tri_count       per edge: popcount(row_i & row_j & bits above j)          -> scan -> ordinals
//   3. tri_keys        per triangle: key at wkey[ordinal]  (s_ik / s_jk come from the compact edge-weight
//                      array through prefix popcounts, not from the 4N^2-byte dense matrix)
//   4. select rounds   exact radix select of the T-th largest key over wkey (<= 3 histogram rounds)
//   5. compact         keys > k*, plus the first (T - #greater) keys == k* by ordinal, in ordinal order
//   6. sort (rocPRIM)  by (~key, position)    7. tri_decode  ordinal -> (i,j,k)
// Everything is integer / bit work except the two fp32 adds of the key; results do not depend on grid size
// or on the order atomics land in (atomics are only used for commutative integer sums, min and max).
#include <cstring>

#include "sc_arith.hpp"
#include "sc_block.hpp"
#include "sc_kernels.hpp"
#include "sc_gramref.hpp"

namespace sc {

__device__ __forceinline__ uint64_t mask_above(int bit) {  // bits strictly above `bit` (0..63)
  return bit == 63 ? 0ull : (~0ull << (bit + 1));
}

// exclusive prefix over the `width`-lane group of a wave (width = 16 or 64); *total = group sum
template <int WIDTH>
__device__ __forceinline__ uint32_t group_exscan(uint32_t v, uint32_t* total) {
  const int gl = threadIdx.x & (WIDTH - 1);
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < WIDTH; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, WIDTH);
    if (gl >= o) inc += t;
  }
  *total = __shfl(inc, WIDTH - 1, WIDTH);
  return inc - v;
}


// Pop up to four set bits of m (lowest first).  nb = how many were popped; b[] are their positions (0 for unused
// slots, so derived indices stay in range).  Lets a lane issue the gathers of four triangles before it consumes the
// first: a lane's triangles would otherwise cost one dependent L2 round trip each.
__device__ __forceinline__ int pop4(uint64_t& m, int b[4]) {
  int nb = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const bool has = m != 0;
    b[q] = has ? __builtin_ctzll(m) : 0;
    m = has ? (m & (m - 1)) : 0;
    nb += has ? 1 : 0;
  }
  return nb;
}

// ------------------------------------------------------------------------------------------------
// 1. edge_fill: one wave per row
// ------------------------------------------------------------------------------------------------
// The edge weight is RECOMPUTED from the two correspondences with stage A's own chain (pair_weight over dist3: the
// same correctly rounded operations on the same operands give the same bits; the squares make the argument order
// irrelevant) instead of being gathered from S: a gather moves a 64-byte sector of the n x n matrix for 4 useful
// bytes (C3: 215 us), the recomputation is ~80 VALU operations on points that sit in L2.
__device__ __forceinline__ uint32_t weight_bin(uint32_t wbits, uint32_t wlo, uint32_t wshift);  // §3b

__global__ __launch_bounds__(256) void edge_fill_kernel(const uint64_t* __restrict__ bits,
                                                        const float* __restrict__ planes, Derived dv,
                                                        int n, int ld, int W,
                                                        const uint64_t* __restrict__ edge_off,
                                                        uint32_t* __restrict__ ei, uint32_t* __restrict__ ej,
                                                        float* __restrict__ es,
                                                        const uint32_t* __restrict__ deg,
                                                        const uint32_t* __restrict__ degp,
                                                        uint32_t* __restrict__ ebase, int ebase_ready,
                                                        uint32_t* __restrict__ ebi,
                                                        uint32_t* __restrict__ ebj, uint64_t cap,
                                                        uint32_t* __restrict__ es_hist) {
  // cap: entries the edge arrays hold.  The host may launch this kernel BEFORE it knows the edge count (into the
  // arrays of the previous call, while it polls for the count); writes beyond cap are dropped and the host re-runs.
  // es_hist (optional): the 256-bin histogram of the edge weights that sizes the heaviest-edge pruning sample (§3b;
  // PR_HCOPIES global copies) — collected here, where the weights are made, instead of by a launch of its own (7 us).
  constexpr int CH = 512;  // column indices staged per wave and chunk
  __shared__ uint32_t l_j[4][CH];
  __shared__ uint32_t l_h[2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (es_hist) {
    l_h[0][threadIdx.x] = 0u; l_h[1][threadIdx.x] = 0u;
    __syncthreads();
  }
  if (i < n) {
  uint64_t base = edge_off[i];
  // CSR base of row i: edge_off[i] - (# bits of row i at or below i) = edge_off[i] - (deg - deg+), modular u32.
  // ebase_ready: the (tiled) scan of deg+ wrote every row's base already, so the base of the OTHER end is one gather;
  // otherwise (small n: a one-block scan, where the extra loads cost more than they save here) it is derived on the fly
  // from three and this wave records its own row's.
  const uint32_t my_base = ebase_ready ? ebase[i] : (uint32_t)base - (deg[i] - degp[i]);
  if (!ebase_ready && lane == 0) ebase[i] = my_base;
  const int w0 = i >> 6;
  const float pix = planes[i], piy = planes[ld + i], piz = planes[2 * ld + i];  // wave-uniform: scalar loads
  const float qix = planes[3 * ld + i], qiy = planes[4 * ld + i], qiz = planes[5 * ld + i];
  const float4* __restrict__ aos4 = reinterpret_cast<const float4*>(planes + 6 * (size_t)ld);
  for (int wb = w0; wb < W; wb += 64) {
    const int w = wb + lane;
    uint64_t v = 0;
    if (w < W) {
      v = bits[(size_t)i * W + w];
      if (w == w0) v &= mask_above(i & 63);
    }
    uint32_t tot;
    const uint32_t r = group_exscan<64>((uint32_t)__popcll(v), &tot);
    for (uint32_t c0 = 0; c0 < tot; c0 += CH) {  // wave-uniform; one chunk unless the row is very dense
      // (1) divergent part, LDS only: the columns of this chunk in ascending order
      uint64_t vv = v;
      uint32_t rr = r - c0;  // modular: positions outside [0, CH) are skipped
      while (vv) {
        const int b = __builtin_ctzll(vv);
        vv &= vv - 1;
        if (rr < (uint32_t)CH) l_j[wave][rr] = (uint32_t)(w * 64 + b);
        rr++;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS is in-order within a wave
      // (2) flat part, one lane per edge: gathers, weight, coalesced stores
      const uint32_t cnt = min((uint32_t)CH, tot - c0);
      for (uint32_t t = lane; t < cnt; t += 64) {
        const uint32_t j = l_j[wave][t];
        const uint64_t e = base + c0 + t;
        if (e >= cap) continue;
        // the other end from the AoS copy behind the planes (8 floats per correspondence): two 16-byte loads
        const float4 a4 = aos4[2 * (size_t)j], b4 = aos4[2 * (size_t)j + 1];
        const float dp = dist3(pix, piy, piz, a4.x, a4.y, a4.z);
        const float dq = dist3(qix, qiy, qiz, a4.w, b4.x, b4.y);
        bool edge;
        const float sw = pair_weight(dp, dq, dv.d_thr, dv.min_len, dv.neg_inv2sig2, edge);
        ei[e] = (uint32_t)i;
        ej[e] = j;
        es[e] = sw;
        if (es_hist) atomicAdd(&l_h[lane & 1][weight_bin(__float_as_uint(sw), 0u, 0u)], 1u);  // integer sums: order-free
        // both CSR bases travel with the edge, so stage B fetches an edge in ONE memory level
        ebi[e] = my_base;
        ebj[e] = ebase_ready ? ebase[j] : (uint32_t)edge_off[j] - (deg[j] - degp[j]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    base += tot;
  }
  }  // i < n
  if (es_hist) {
    __syncthreads();
    const uint32_t v = l_h[0][threadIdx.x] + l_h[1][threadIdx.x];
    if (v) atomicAdd(&es_hist[(size_t)(blockIdx.x & (PR_HCOPIES - 1)) * 256 + threadIdx.x], v);
  }
}

void launch_edge_fill(const Graph& g, const Points& pts, const Derived& dv, const uint64_t* edge_off, uint32_t* ei,
                      uint32_t* ej, float* es, uint32_t* ebase, bool ebase_ready, uint32_t* ebi, uint32_t* ebj,
                      uint64_t cap, uint32_t* es_hist, hipStream_t st) {
  hipLaunchKernelGGL(edge_fill_kernel, dim3((g.n + 3) / 4), dim3(256), 0, st, g.bits, pts.planes, dv, g.n, g.ld, g.W,
                     edge_off, ei, ej, es, g.deg, g.degp, ebase, ebase_ready ? 1 : 0, ebi, ebj, cap, es_hist);
}

// ------------------------------------------------------------------------------------------------
// 2. tri_count: 16 lanes per edge
// ------------------------------------------------------------------------------------------------
// lanes per edge: Tuning::tg_count / tg_keys / tg_sample (4|8|16|32|64); 8 measured best on C2

constexpr int TK_MAX_BLOCKS = 4096;  // bounds the per-block min/max arrays

// mbits: the bit matrix that decides MEMBERSHIP (the full adjacency, or the pruned "strong" upper-triangle matrix);
// smin != nullptr: edges with es[e] < *smin are outside the pruned graph and count 0 without touching any row.
// EB edges per group are processed together: every dependent memory level (edge record -> rows -> ...) is then paid
// once per batch instead of once per edge.  These kernels are latency-bound (PMC: > 60 % of wave cycles waiting), so
// the batch is worth ~EB in time until the L2 request queues fill.
constexpr int EB = 4;

template <int TG>
__global__ __launch_bounds__(256) void tri_count_kernel(const uint64_t* __restrict__ mbits, int W,
                                                        const uint32_t* __restrict__ ei,
                                                        const uint32_t* __restrict__ ej,
                                                        const float* __restrict__ es,
                                                        const float* __restrict__ smin, uint64_t E,
                                                        uint32_t* __restrict__ tcnt,
                                                        const uint64_t* __restrict__ own) {
  // own (optional): [lo, hi) of the edges this rank enumerates (sharded stage B); the others count 0
  const int gl = threadIdx.x & (TG - 1);
  const float s_floor = smin ? *smin : -1.0f;
  const uint64_t own_lo = own ? own[0] : 0ull, own_hi = own ? own[1] : E;
  const uint64_t groups = (uint64_t)gridDim.x * (256 / TG);
  const uint64_t g0 = (uint64_t)blockIdx.x * (256 / TG) + (threadIdx.x / TG);
  for (uint64_t e0 = g0 * EB; e0 < E; e0 += groups * EB) {  // a group owns EB consecutive edges per trip
    uint32_t rowi[EB], rowj[EB], c[EB];
    int w0[EB], jb[EB];
    bool on[EB];
#pragma unroll
    for (int q = 0; q < EB; q++) {  // level 1: the edge records of the whole batch
      const uint64_t e = e0 + q;
      on[q] = e < E;
      const uint64_t ec = on[q] ? e : 0;
      const float s = es[ec];
      const uint32_t i = ei[ec], j = ej[ec];
      on[q] = on[q] && (s >= s_floor) && e >= own_lo && e < own_hi;
      rowi[q] = i * (uint32_t)W; rowj[q] = j * (uint32_t)W;
      w0[q] = (int)(j >> 6); jb[q] = (int)(j & 63);
      c[q] = 0;
    }
    int wmin = W;
#pragma unroll
    for (int q = 0; q < EB; q++) wmin = on[q] ? min(wmin, w0[q]) : wmin;
    for (int wb = wmin; wb < W; wb += TG) {  // level 2..: one round of words for all EB edges at a time
      const int w = wb + gl;
      uint64_t a[EB], bq[EB];
#pragma unroll
      for (int q = 0; q < EB; q++) {
        const bool live = on[q] && w >= w0[q] && w < W;
        a[q] = live ? mbits[rowi[q] + w] : 0ull;
        bq[q] = live ? mbits[rowj[q] + w] : 0ull;
      }
#pragma unroll
      for (int q = 0; q < EB; q++) {
        uint64_t m = a[q] & bq[q];
        if (w == w0[q]) m &= mask_above(jb[q]);
        c[q] += (uint32_t)__popcll(m);
      }
    }
#pragma unroll
    for (int q = 0; q < EB; q++) {
#pragma unroll
      for (int o = TG / 2; o > 0; o >>= 1) c[q] += __shfl_xor(c[q], o, TG);
      if (gl == 0 && e0 + q < E) tcnt[e0 + q] = c[q];
    }
  }
}

void launch_tri_count(const Graph& g, const uint64_t* mbits, const float* es, const float* smin, const uint32_t* ei,
                      const uint32_t* ej, uint64_t E, uint32_t* tcnt, const uint64_t* own, const Tuning& tn,
                      hipStream_t st) {
  if (E == 0) return;
  const int tg = tn.tg_count;
  const uint64_t per = (uint64_t)(256 / tg) * EB;
  uint64_t nb = (E + per - 1) / per;
  if (nb > 4096) nb = 4096;
#define SC_LAUNCH_COUNT(TGV) hipLaunchKernelGGL(tri_count_kernel<TGV>, dim3((unsigned)nb), dim3(256), 0, st, mbits, g.W, ei, ej, es, smin, E, tcnt, own)
  if (tg == 4) SC_LAUNCH_COUNT(4); else if (tg == 8) SC_LAUNCH_COUNT(8); else if (tg == 32) SC_LAUNCH_COUNT(32); else SC_LAUNCH_COUNT(16);
#undef SC_LAUNCH_COUNT
}

// ------------------------------------------------------------------------------------------------
// 3. tri_keys
// ------------------------------------------------------------------------------------------------

// bits / wpre / ebase: full adjacency + its word-prefix popcounts + per-row CSR bases: the index of edge (v,k) in the
// edge arrays is ebase[v] + wpre[v][k/64] + popc(bits[v][k/64] & below k) — an O(1) lookup, no running prefix.
// mbits: membership matrix (== bits when nothing is pruned); smin: strong-edge threshold or nullptr.
// Per round only the member words are ANDed; the one remaining group scan (triangle rank inside the edge, for the
// ordinal) is skipped when the whole group found nothing — the common case in the pruned graph.
template <int TG>
__global__ __launch_bounds__(256) void tri_keys_kernel(const uint64_t* __restrict__ bits,
                                                       const uint64_t* __restrict__ mbits,
                                                       const float* __restrict__ smin, int W,
                                                       const uint32_t* __restrict__ deg,
                                                       const uint32_t* __restrict__ wpre,
                                                       const uint32_t* __restrict__ ebase,
                                                       const uint32_t* __restrict__ ei,
                                                       const uint32_t* __restrict__ ej,
                                                       const float* __restrict__ es,
                                                       const uint64_t* __restrict__ toff, uint64_t E,
                                                       int rank_mode, uint32_t* __restrict__ wkey,
                                                       uint2* __restrict__ kcol,
                                                       uint32_t* __restrict__ blk_min,
                                                       uint32_t* __restrict__ blk_max,
                                                       const uint64_t* __restrict__ own) {
  __shared__ uint32_t lmin[4], lmax[4];
  const int gl = threadIdx.x & (TG - 1);
  const uint64_t own_lo = own ? own[0] : 0ull, own_hi = own ? own[1] : E;
  const uint64_t gmask = (TG == 64) ? ~0ull : (((1ull << TG) - 1ull) << ((threadIdx.x & 63) & ~(TG - 1)));
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  const uint64_t groups = (uint64_t)gridDim.x * (256 / TG);
  const bool pruned = (mbits != bits);
  const float s_floor = smin ? *smin : -1.0f;
  for (uint64_t e = (uint64_t)blockIdx.x * (256 / TG) + (threadIdx.x / TG); e < E; e += groups) {
    const float s_ij = es[e];
    if (s_ij < s_floor || e < own_lo || e >= own_hi) continue;  // group-uniform: outside the pruned graph / not this rank's
    const uint32_t i = ei[e], j = ej[e];
    const uint32_t rowi = i * (uint32_t)W, rowj = j * (uint32_t)W;  // word offsets: n * W < 2^32
    const int w0 = j >> 6;
    const uint32_t dsum_ij = (rank_mode == 0) ? 0u : deg[i] + deg[j];
    const uint32_t ebi = ebase[i], ebj = ebase[j];
    uint32_t out = (uint32_t)0;  // triangle rank inside the edge
    const uint64_t out0 = toff[e];
    const int rounds = (W - w0 + TG - 1) / TG;
    for (int it = 0; it < rounds; it++) {
      const int w = w0 + it * TG + gl;
      uint64_t m = 0;
      if (w < W) {
        m = mbits[rowi + w] & mbits[rowj + w];
        if (w == w0) m &= mask_above(j & 63);
      }
      const uint64_t any = __ballot(m != 0) & gmask;
      if (any == 0) continue;  // group-uniform
      uint32_t tm;
      uint32_t pm = out + group_exscan<TG>((uint32_t)__popcll(m), &tm);
      out += tm;
      if (m) {
        // full-row words and prefixes only where a triangle was found
        const uint64_t ai = pruned ? bits[rowi + w] : 0ull, aj = pruned ? bits[rowj + w] : 0ull;
        const uint64_t fi = pruned ? ai : mbits[rowi + w], fj = pruned ? aj : mbits[rowj + w];
        const uint32_t pi = ebi + wpre[rowi + w], pj = ebj + wpre[rowj + w];
        while (m) {
          int b[4];
          const int nbits = pop4(m, b);
          uint32_t key[4];
          if (rank_mode == 0) {
            float s_ik[4], s_jk[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {  // all eight gathers are issued before the first use
              const uint64_t below = (1ull << b[q]) - 1ull;
              const bool live = q < nbits;  // idle slots read es[0]: their modular index may be out of range
              const uint32_t xi = live ? pi + (uint32_t)__popcll(fi & below) : 0u;
              const uint32_t xj = live ? pj + (uint32_t)__popcll(fj & below) : 0u;
              s_ik[q] = es[xi];
              s_jk[q] = es[xj];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) key[q] = __float_as_uint((s_ij + s_ik[q]) + s_jk[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) key[q] = dsum_ij + deg[w * 64 + b[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (q < nbits) {
              kcol[out0 + pm] = make_uint2((uint32_t)(w * 64 + b[q]), (uint32_t)e);  // third vertex + edge: tri_decode
              wkey[out0 + pm++] = key[q];
              kmin = min(kmin, key[q]);
              kmax = max(kmax, key[q]);
            }
          }
        }
      }
    }
  }
  // block min/max -> one plain store per block (no same-address atomics: they serialise at ~12 ns each)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor(kmin, o));
    kmax = max(kmax, (uint32_t)__shfl_xor(kmax, o));
  }
  if ((threadIdx.x & 63) == 0) { lmin[threadIdx.x >> 6] = kmin; lmax[threadIdx.x >> 6] = kmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_min[blockIdx.x] = min(min(lmin[0], lmin[1]), min(lmin[2], lmin[3]));
    blk_max[blockIdx.x] = max(max(lmax[0], lmax[1]), max(lmax[2], lmax[3]));
  }
}

__global__ __launch_bounds__(1024) void key_range_kernel(const uint32_t* __restrict__ blk_min,
                                                         const uint32_t* __restrict__ blk_max, int nb,
                                                         SelectState* __restrict__ sel, uint64_t want) {
  __shared__ uint32_t lmin[16], lmax[16];
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  for (int b = threadIdx.x; b < nb; b += 1024) { kmin = min(kmin, blk_min[b]); kmax = max(kmax, blk_max[b]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor(kmin, o));
    kmax = max(kmax, (uint32_t)__shfl_xor(kmax, o));
  }
  if ((threadIdx.x & 63) == 0) { lmin[threadIdx.x >> 6] = kmin; lmax[threadIdx.x >> 6] = kmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) { kmin = min(kmin, lmin[w]); kmax = max(kmax, lmax[w]); }
    sel->kmin = kmin; sel->kmax = kmax;
    sel->st[0] = SelSnap{0u, 0u, 0u, 0u, want, 0ull, 0ull, 0ull};  // (a speculative key pass may have preset a window)
    sel->want_req = 0;
  }
}

size_t tri_keys_blocks(uint64_t E, int tg) {
  const uint64_t per = 256 / tg;
  const uint64_t nb = (E + per - 1) / per;
  return (size_t)(nb < (uint64_t)TK_MAX_BLOCKS ? nb : (uint64_t)TK_MAX_BLOCKS);
}

void launch_tri_keys(const Graph& g, const uint64_t* mbits, const float* smin, const uint32_t* ebase,
                     const uint32_t* ei, const uint32_t* ej, const float* es, const uint64_t* toff, uint64_t E,
                     int rank_mode, uint32_t* wkey, uint2* kcol, uint32_t* blk_minmax, SelectState* s,
                     uint64_t want, const uint64_t* own, const Tuning& tn, hipStream_t st) {
  if (E == 0) return;
  const int tg = tn.tg_keys;
  const int nb = (int)tri_keys_blocks(E, tg);
#define SC_LAUNCH_KEYS(TGV) hipLaunchKernelGGL(tri_keys_kernel<TGV>, dim3(nb), dim3(256), 0, st, g.bits, mbits, smin, g.W, g.deg, g.wpre, ebase, ei, ej, es, toff, E, rank_mode, wkey, kcol, blk_minmax, blk_minmax + TK_MAX_BLOCKS, own)
  if (tg == 4) SC_LAUNCH_KEYS(4); else if (tg == 8) SC_LAUNCH_KEYS(8); else if (tg == 32) SC_LAUNCH_KEYS(32); else if (tg == 64) SC_LAUNCH_KEYS(64); else SC_LAUNCH_KEYS(16);
#undef SC_LAUNCH_KEYS
  hipLaunchKernelGGL(key_range_kernel, dim3(1), dim3(1024), 0, st, blk_minmax, blk_minmax + TK_MAX_BLOCKS, nb, s, want);
}

// ------------------------------------------------------------------------------------------------
// 2b/3'. event list: count and keys without enumerating twice
//
// tri_count and tri_keys used to AND the same pairs of bit rows, and tri_keys paid three dependent memory levels per
// wave-round (rows -> prefix words -> edge weights).  Now the counting pass also records every non-zero member word
// as an EVENT {member word m, word offsets of both rows, CSR bases of both ends, edge, rank of its first triangle
// inside the edge}.  Events are staged per WAVE in LDS and appended to one of EV_SHARDS global regions with a single
// atomic per flush (one counter per region: same-address RETURNING atomics serialise at ~170 ns each).  After the scan
// of the per-edge counts, tri_keys_events_kernel runs one lane per event: all its gathers are independent, and a
// triangle's key lands at its ordinal toff[e] + rank.  Event order is arbitrary; the result is not.
// If a region overflows, a flag reaches the host with the triangle count and the call falls back to the row-walking
// tri_keys_kernel (and doubles the event capacity for the next call).
// ------------------------------------------------------------------------------------------------
constexpr int EV_SHARDS = 1024;  // ~26 k flushes per call on C2: with 256 counters the returning atomics serialise (7.7 us)

// Staging is per WAVE (EVW records of LDS each).  The rounds loop is wave-uniform: it runs to the largest round count
// among the wave's groups and idle groups contribute empty words, so (a) events are appended with __ballot / mbcnt —
// no atomics, (b) the flush decision is taken by the whole wave after any round — a round adds at most 64 records, so
// the segment can never overflow, whatever the row width, and (c) no workgroup barrier is needed.  One global atomic
// per flush, on one of EV_SHARDS counters.
// Shapes measured and rejected (C2 unless noted): workgroup staging with two barriers per trip (48 us, and it overflows
// into per-event global atomics at W = 313: 4.7 ms on C3), per-wave staging with a global-atomic overflow path
// (235 us), per-group staging (118 us), a one-round-per-iteration state machine with software prefetch (98 us).
template <int TG>
__global__ __launch_bounds__(256) void tri_count_events_kernel(const uint64_t* __restrict__ mbits, int W,
                                                               const uint32_t* __restrict__ deg,
                                                               const uint32_t* __restrict__ ebi,
                                                               const uint32_t* __restrict__ ebj,
                                                               const uint32_t* __restrict__ ei,
                                                               const uint32_t* __restrict__ ej,
                                                               StrongList sl, int rank_mode,
                                                               uint32_t* __restrict__ tcnt, EventList ev,
                                                               const uint64_t* __restrict__ own,
                                                               const uint32_t* __restrict__ ebase, GramRefJob ref) {
  // ref (ref.out != null): workgroup 0 is a RIDER — it does none of this kernel's work but votes for the reference frame of
  // stage C2's Gram filter among the candidate triangles the estimating sample left behind (sc_gramref.hpp): ~10 us of one
  // workgroup's latency that would otherwise stand between the selection and the Kabsch launch, hidden under this launch
  const uint32_t rider = ref.out ? 1u : 0u;
  const uint32_t bid = blockIdx.x - rider, nblk = gridDim.x - rider;
  // ebase (with ebi == ebj == nullptr): the per-row CSR bases are looked up (edge_build_kernel does not write them per edge)
  // own (optional, sharded stage B): [lo, hi) of the edges this rank enumerates; the strong list holds every rank's, the
  // others count 0 here (their tcnt entry is written too: the scan that follows reads zeros outside the range)
  constexpr int EVW = 192;  // records per wave segment: flush above 128, a round adds <= 64
  __shared__ uint64_t l_m[4 * EVW];
  __shared__ uint32_t l_wi[4 * EVW], l_wj[4 * EVW], l_a[4 * EVW], l_b[4 * EVW], l_e[4 * EVW], l_rb[4 * EVW];
  __shared__ uint32_t l_pre[ST_SHARDS + 1];  // exclusive prefix of the strong-list region fills
  static_assert(sizeof(l_m) >= GX_REF_LDS_WORDS * sizeof(float), "the rider borrows the event staging area");
  if (rider && blockIdx.x == 0) { gram_ref_block(ref, reinterpret_cast<float*>(l_m)); return; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = threadIdx.x & (TG - 1);
  static_assert(ST_SHARDS == 256, "one region per thread");
  {
    __shared__ uint64_t plds[8];
    const uint64_t v = sl.fill[threadIdx.x];
    uint64_t tot;
    l_pre[threadIdx.x] = (uint32_t)block_exscan_u64(v, plds, &tot);
    if (threadIdx.x == 0) l_pre[ST_SHARDS] = (uint32_t)tot;
  }
  __syncthreads();
  const uint64_t S = l_pre[ST_SHARDS];  // strong edges (only they can carry a triangle of the pruned graph)
  // the part of the flat list this rank walks: everything, or — regions being contiguous edge ranges — the regions its own
  // edge range touches (the others' tcnt entries were zeroed by the pruning kernel)
  uint64_t x0 = 0, x1 = S;
  if (own && sl.region_blocks) {
    const uint64_t per_region = (uint64_t)sl.region_blocks * 256;
    if (own[1] > own[0]) {
      const uint64_t r_lo = min(own[0] / per_region, (uint64_t)(ST_SHARDS - 1));
      const uint64_t r_hi = min((own[1] - 1) / per_region, (uint64_t)(ST_SHARDS - 1));
      x0 = l_pre[r_lo]; x1 = l_pre[r_hi + 1];
    } else {
      x1 = 0;
    }
  }
  const uint64_t groups = (uint64_t)nblk * (256 / TG);
  const uint64_t g0 = (uint64_t)bid * (256 / TG) + (threadIdx.x / TG);
  // region of this wave's NEXT ticket: it moves on by one with every ticket, so a wave with many events spreads them over
  // the regions (r03: every wave kept one region and a region overflowed in one C4 call in three at 2.2 records of capacity
  // per event — the call then fell back to the row-walking key kernel)
  uint32_t shard = (bid * 4 + wave) & (EV_SHARDS - 1);
  const uint64_t trips = (x1 - x0 + groups - 1) / groups;  // the same for every lane of the wave
  const int wbase = wave * EVW;
  uint32_t scnt = 0;  // records staged by this wave (wave-uniform)
  auto flush = [&]() {  // wave-uniform
    // A ticket that does not fit whole leaves no hole: the part inside the region is written (the key kernel reads
    // min(fill, capacity) records of a region), the rest moves on to the next region.  So the list overflows only when the
    // regions are full TOGETHER — a single region filling early (the strong list keeps the edges of a row together, so a run
    // of waves inside the inlier clique stages several times the mean) costs a second ticket, not the whole call's event path.
    uint32_t start = 0;  // records [start, scnt) are still to be placed
    for (int tries = 0; tries < EV_SHARDS && start < scnt; tries++) {
      const uint32_t cnt = scnt - start;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&ev.fill[shard], cnt);
      base = __builtin_amdgcn_readfirstlane(base);
      const uint32_t fit = (uint64_t)base >= ev.shard_cap ? 0u : (uint32_t)min((uint64_t)cnt, ev.shard_cap - (uint64_t)base);
      for (uint32_t k = lane; k < fit; k += 64) {
        const uint64_t o = (uint64_t)shard * ev.shard_cap + (uint64_t)base + k;
        const int q = wbase + (int)(start + k);
        ev.m[o] = l_m[q]; ev.wi[o] = l_wi[q]; ev.wj[o] = l_wj[q]; ev.a[o] = l_a[q]; ev.b[o] = l_b[q];
        ev.e[o] = l_e[q]; ev.rb[o] = l_rb[q];
      }
      start += fit;
      shard = (shard + 1u) & (EV_SHARDS - 1);
    }
    if (start < scnt && lane == 0) *ev.overflow = 1u;
    scnt = 0;
  };
  for (uint64_t trip = 0; trip < trips; trip++) {
    const uint64_t x = x0 + g0 + trip * groups;  // position in the flat strong list
    const bool on = x < x1;                      // group-uniform
    uint32_t e = 0, rowi = 0, rowj = 0, fa = 0, fb = 0, c = 0;
    int w0 = 0, jbit = 0, rounds = 0;
    if (on) {
      int r = 0;  // region of x: the largest r with l_pre[r] <= x
#pragma unroll
      for (int step = ST_SHARDS / 2; step > 0; step >>= 1) r += (l_pre[r + step] <= (uint32_t)x) ? step : 0;
      e = sl.list[(uint64_t)r * sl.cap + ((uint32_t)x - l_pre[r])];
      if (!own || ((uint64_t)e >= own[0] && (uint64_t)e < own[1])) {
        const uint32_t i = ei[e], j = ej[e];
        rowi = i * (uint32_t)W; rowj = j * (uint32_t)W;
        fa = (rank_mode == 0) ? (ebi ? ebi[e] : ebase[i]) : deg[i] + deg[j];  // weight mode: CSR bases; degree mode: the degree sum
        fb = (rank_mode == 0) ? (ebj ? ebj[e] : ebase[j]) : 0u;
        w0 = (int)(j >> 6); jbit = (int)(j & 63);
        rounds = (W - w0 + TG - 1) / TG;
      }
    }
    int wave_rounds = rounds;  // max over the wave: the loop below is wave-uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_rounds = max(wave_rounds, __shfl_xor(wave_rounds, o));
    auto take = [&](uint64_t m, int w) {  // one round's words: count, stage the non-zero ones (wave-uniform control flow)
      const uint64_t bal = __ballot(m != 0);
      if (bal != 0) {
        uint32_t tm;
        const uint32_t rb = c + group_exscan<TG>((uint32_t)__popcll(m), &tm);
        c += tm;
        if (m) {
          const int q = wbase + (int)scnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          l_m[q] = m; l_wi[q] = rowi + w; l_wj[q] = rowj + w; l_a[q] = fa; l_b[q] = fb; l_e[q] = e; l_rb[q] = rb;
        }
        scnt += (uint32_t)__popcll(bal);
        if (scnt > (uint32_t)(EVW - 64)) flush();
      }
    };
    // SEVERAL rounds' row words are loaded before the first is looked at (r05, by time stamps inside the workgroups: a round is one
    // L2 round trip, ~0.45 us, and an edge's 6 - 19 rounds were most of a workgroup's 9 - 14 us)
    constexpr int RU = 2;  // rounds whose words are in flight together (4: no better at C2 and C4, + 6 us at C3)
    for (int it = 0; it < wave_rounds; it += RU) {
      uint64_t mi[RU], mj[RU];
#pragma unroll
      for (int u = 0; u < RU; u++) {
        const int w = w0 + (it + u) * TG + gl;
        const bool in = it + u < rounds && w < W;
        mi[u] = in ? mbits[rowi + w] : 0ull;
        mj[u] = in ? mbits[rowj + w] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < RU; u++) {
        if (it + u >= wave_rounds) break;  // (wave-uniform)
        const int w = w0 + (it + u) * TG + gl;
        uint64_t m = mi[u] & mj[u];
        if (w == w0) m &= mask_above(jbit);
        take(m, w);
      }
    }
    if (gl == 0 && on) tcnt[e] = c;  // weak edges were zeroed by prune_bits_kernel
  }
  if (scnt) flush();
}

// one lane per event
__global__ __launch_bounds__(256) void tri_keys_events_kernel(const uint64_t* __restrict__ bits,
                                                              const uint32_t* __restrict__ wpre,
                                                              const uint32_t* __restrict__ deg,
                                                              const float* __restrict__ es,
                                                              const uint64_t* __restrict__ toff, int rank_mode,
                                                              EventList ev, uint32_t* __restrict__ wkey,
                                                              uint2* __restrict__ kcol,
                                                              uint32_t* __restrict__ blk_min,
                                                              uint32_t* __restrict__ blk_max,
                                                              SelectState* __restrict__ preset,
                                                              const uint32_t* __restrict__ klb, uint64_t want,
                                                              uint64_t E, uint64_t cap, int check_bound) {
  // cap: entries wkey / kcol hold.  The host may launch this kernel BEFORE it knows the triangle count (into the
  // arrays of the previous call, while it polls for the count): writes beyond cap are dropped and the host re-runs.
  // For the same reason `want` is clipped here to the count the scan left in toff[E].
  __shared__ uint32_t lmin[4], lmax[4];
  __shared__ uint64_t pre[EV_SHARDS + 1];
  // Weight keys of a graph whose edges all weigh >= 2/3 live in one binade, [2.0, 3.0]: the select window is known
  // before a single key exists — [certified bound (or 2.0), 3.0] — so no key-range pass, and two 12-bit rounds
  // always resolve it.  (Keys below a certified bound cannot be among the `want` largest: sc_tri.hip 3b.)
  if (preset && blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t lo = *klb ? *klb : 0x40000000u, hi = 0x40400000u;  // 2.0f, 3.0f
    const uint32_t range_m1 = hi - lo;
    preset->kmin = lo; preset->kmax = hi;
    preset->st[0] = SelSnap{lo, range_m1 == 0 ? 0u : (uint32_t)(32 - __builtin_clz(range_m1)), 1u, 0u, min(want, (uint64_t)toff[E]), 0ull, 0ull, 0ull};
    // a pruning bound promises `want` keys at or above it — a certified one by construction, an ESTIMATED one (3c) unless
    // it was set too high: whoever resolves the select's first round compares (the UNclipped want: a pruned graph with fewer triangles than
    // that proves nothing about the full one)
    preset->want_req = (check_bound && *klb) ? want : 0ull;
  }  // exclusive prefix of the region fills: one flat index space over all events
  {
    constexpr int PT = EV_SHARDS / 256;  // regions per thread
    __shared__ uint64_t plds[8];
    uint64_t f[PT], mine = 0;
#pragma unroll
    for (int k = 0; k < PT; k++) { f[k] = min((uint64_t)ev.fill[threadIdx.x * PT + k], ev.shard_cap); mine += f[k]; }
    uint64_t tot;
    uint64_t run = block_exscan_u64(mine, plds, &tot);
#pragma unroll
    for (int k = 0; k < PT; k++) { pre[threadIdx.x * PT + k] = run; run += f[k]; }
    if (threadIdx.x == 0) pre[EV_SHARDS] = tot;
    __syncthreads();
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  const uint64_t nthreads = (uint64_t)gridDim.x * 256, total = pre[EV_SHARDS];
  {
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += nthreads) {
      int sh = 0;  // largest sh with pre[sh] <= x
#pragma unroll
      for (int step = EV_SHARDS / 2; step > 0; step >>= 1) sh += (pre[sh + step] <= x) ? step : 0;
      const uint64_t o = (uint64_t)sh * ev.shard_cap + (x - pre[sh]);
      uint64_t m = ev.m[o];
      const uint32_t wi = ev.wi[o], wj = ev.wj[o], fa = ev.a[o], e = ev.e[o];
      uint64_t out = toff[e] + ev.rb[o];
      const uint32_t kbase = (uint32_t)(wi % (uint32_t)ev.W) * 64u;  // column index of bit 0 of this word
      if (rank_mode == 0) {
        const uint32_t fb = ev.b[o];
        const uint64_t fi = bits[wi], fj = bits[wj];  // full-graph words: ranks in the CSR edge arrays
        const uint32_t pi = fa + wpre[wi], pj = fb + wpre[wj];
        const float s_ij = es[e];
        while (m) {
          int b[4];
          const int nbits = pop4(m, b);
          float s_ik[4], s_jk[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint64_t below = (1ull << b[q]) - 1ull;
            const bool live = q < nbits;  // idle slots read es[0]
            s_ik[q] = es[live ? pi + (uint32_t)__popcll(fi & below) : 0u];
            s_jk[q] = es[live ? pj + (uint32_t)__popcll(fj & below) : 0u];
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (q < nbits) {
              const uint32_t key = __float_as_uint((s_ij + s_ik[q]) + s_jk[q]);
              if (out < cap) { kcol[out] = make_uint2(kbase + (uint32_t)b[q], e); wkey[out] = key; }
              out++;
              kmin = min(kmin, key);
              kmax = max(kmax, key);
            }
          }
        }
      } else {
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const uint32_t key = fa + deg[kbase + b];
          if (out < cap) { kcol[out] = make_uint2(kbase + (uint32_t)b, e); wkey[out] = key; }
          out++;
          kmin = min(kmin, key);
          kmax = max(kmax, key);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor(kmin, o));
    kmax = max(kmax, (uint32_t)__shfl_xor(kmax, o));
  }
  if ((threadIdx.x & 63) == 0) { lmin[threadIdx.x >> 6] = kmin; lmax[threadIdx.x >> 6] = kmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_min[blockIdx.x] = min(min(lmin[0], lmin[1]), min(lmin[2], lmin[3]));
    blk_max[blockIdx.x] = max(max(lmax[0], lmax[1]), max(lmax[2], lmax[3]));
  }
}

size_t event_bytes(uint64_t capacity) { return (size_t)capacity * 32 + EV_SHARDS * 4 + 64; }

EventList event_list(void* buf, uint64_t capacity, int W, uint32_t* fill, uint32_t* overflow_host) {
  EventList ev;
  const uint64_t cap = capacity / EV_SHARDS * EV_SHARDS;
  unsigned char* p = static_cast<unsigned char*>(buf);
  ev.m = reinterpret_cast<uint64_t*>(p); p += cap * 8;
  ev.wi = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.wj = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.a = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.b = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.e = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.rb = reinterpret_cast<uint32_t*>(p); p += cap * 4;
  ev.fill = fill;  // EV_SHARDS zeroed counters (control block)
  ev.shard_cap = cap / EV_SHARDS;
  ev.overflow = overflow_host;
  ev.W = W;
  return ev;
}

void launch_tri_count_events(const Graph& g, const uint64_t* mbits, const StrongList& sl, const uint32_t* ebi,
                             const uint32_t* ebj, const uint32_t* ei, const uint32_t* ej, uint64_t E, int rank_mode,
                             uint32_t* tcnt, const EventList& ev, const Tuning& tn, hipStream_t st, const uint64_t* own,
                             const uint32_t* ebase, const GramRefJob* ref) {
  if (E == 0) return;
  // lanes per edge: a lane walks (W - j / 64) / TG words one dependent round after the other, so wide rows want wide
  // groups (Tuning::tg_events forces one; measured r02: see DESIGN.md)
  // r04c, on the pruned graph (a third of the triangles of r03's: shorter walks, more edges per wave pay): 4 lanes between 65 and 128
  // words — C2 (79): stage B's bracket 101.5 -> 99.4 us in eight alternating pairs of runs, C4 (79): 138.6 vs 138.5 — but not
  // below: C1 (32 words) 68.3 -> 69.7 us with 4
  int tg = tn.tg_events ? tn.tg_events : (g.W <= 64 ? 8 : (g.W <= 128 ? 4 : (g.W <= 512 ? 16 : 32)));
  const uint64_t per = 256 / tg;
  // the strong edges are a fraction of E that only the device knows (20 - 45 % on C1 .. C4): size the grid for ~E/3
  uint64_t nb = (E / 3 + per - 1) / per;
  if (nb < 256) nb = 256;
  if (nb > 4096) nb = 4096;
  if (tn.cnt_blocks >= 1 && tn.cnt_blocks <= 65535) nb = tn.cnt_blocks;
  const GramRefJob rj = ref ? *ref : GramRefJob{};
  if (rj.out) nb++;  // (workgroup 0 is the rider)
#define SC_LAUNCH_CE(TGV) hipLaunchKernelGGL(tri_count_events_kernel<TGV>, dim3((unsigned)nb), dim3(256), 0, st, mbits, g.W, g.deg, ebi, ebj, ei, ej, sl, rank_mode, tcnt, ev, own, ebase, rj)
  if (tg == 4) SC_LAUNCH_CE(4); else if (tg == 16) SC_LAUNCH_CE(16); else if (tg == 32) SC_LAUNCH_CE(32); else if (tg == 64) SC_LAUNCH_CE(64); else SC_LAUNCH_CE(8);
#undef SC_LAUNCH_CE
}

void launch_tri_keys_events(const Graph& g, const float* es, const uint64_t* toff, int rank_mode,
                            const EventList& ev, uint32_t* wkey, uint2* kcol, uint32_t* blk_minmax,
                            SelectState* s, uint64_t want, const uint32_t* klb, uint64_t E, uint64_t cap,
                            const Tuning& tn, hipStream_t st, bool check_bound) {
  int nb = 2048;
  if (tn.keys_blocks >= 1 && tn.keys_blocks <= (uint32_t)TK_MAX_BLOCKS) nb = (int)tn.keys_blocks;
  hipLaunchKernelGGL(tri_keys_events_kernel, dim3(nb), dim3(256), 0, st, g.bits, g.wpre, g.deg, es, toff, rank_mode,
                     ev, wkey, kcol, blk_minmax, blk_minmax + TK_MAX_BLOCKS, klb ? s : (SelectState*)nullptr, klb, want, E, cap,
                     check_bound ? 1 : 0);
  if (!klb)  // no a-priori window: the key range comes from the per-block extremes
    hipLaunchKernelGGL(key_range_kernel, dim3(1), dim3(1024), 0, st, blk_minmax, blk_minmax + TK_MAX_BLOCKS, nb, s, want);
}

// ------------------------------------------------------------------------------------------------
// 3b. certified pruning (weight ranking only)
//
// Claim: let LB be any value such that at least T triangles have key >= LB.  Then the top-T list of the full graph
// equals the top-T list of the subgraph of "strong" edges, s >= smin := LB - 2 - 1e-6.
// Proof: w = fl(fl(a+b)+c) with a,b,c in (0,1] satisfies w <= a+b+c + 5*2^-24, so a triangle with w >= LB has every
// edge >= LB - 2 - 3e-7 > smin: all triangles with key >= LB live in the strong subgraph, there are >= T of them, and
// every triangle outside it has key < LB, i.e. strictly below at least T others — it cannot enter the top-T, ties
// included.  Ordinals keep their relative order (same CSR edge order, same ascending k), so the tie-break is unchanged.
// LB comes from a SAMPLE: the triangles of every R-th edge are enumerated once, their keys go into a 256-bin
// histogram over [klo, khi]; walking it from the top to the first bin where the count reaches T gives a bin whose
// lower edge is a valid LB (>= T genuine triangles lie at or above it).  Bin 0 collects everything below klo, so a
// crossing in bin 0 certifies nothing and disables the pruning (smin = -1).
// ------------------------------------------------------------------------------------------------
constexpr int PR_BINS = 256;   // coarse is enough: LB only needs to be a valid, reasonably tight lower bound
constexpr int PR_COPIES = 16;  // one private copy per lane-in-group: same-bin hits land on different LDS words
constexpr int EST_COPIES = 4;  // the estimating sample (one lane per edge, ~200 hits per workgroup): fewer copies to clear and to add up

__device__ __forceinline__ uint32_t prune_bin(uint32_t key, uint32_t klo, uint32_t shift) {
  if (key <= klo) return 0u;
  const uint32_t b = (key - klo) >> shift;
  return b < (uint32_t)PR_BINS ? b : (uint32_t)(PR_BINS - 1);
}

// The ESTIMATING sample (3c) bins by d = 3.0 - key instead — exact in fp32 for keys in [2, 3] — LOGARITHMICALLY, 16 bins per
// octave of d over [2^-17, 2^-1): the top-T keys crowd against 3.0 (three weights near 1), where linear bins of 0.002
// cannot tell the key of rank T from the key of rank 5 T; a bin 4.4 % wide in d is ~13 % wide in rank.  Bin 0: d >= 0.5
// (nothing can be said), bin 255: d < 2^-17 (1 + 1/16).  Selected by shift == PR_LOGBINS.
constexpr uint32_t PR_LOGBINS = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t est_bin(uint32_t key) {
  const float d = 3.0f - __uint_as_float(key);
  if (!(d > 0.0f)) return (uint32_t)(PR_BINS - 1);
  const uint32_t u = __float_as_uint(d) >> 19;  // exponent and four mantissa bits
  return u >= 2016u ? 0u : (u <= 1761u ? (uint32_t)(PR_BINS - 1) : 2016u - u);
}
// a value every key of bin b (>= 1) lies strictly ABOVE: 3.0 - (upper edge of the bin's d interval), exact in fp32
__device__ __forceinline__ float est_bin_floor(uint32_t b) { return 3.0f - __uint_as_float((2017u - b) << 19); }

// Which edges are sampled.  Any subset of genuine triangles certifies a bound; the question is which subset certifies
// a TIGHT one for the fewest key evaluations.
//   TOP = false: every stride-th edge (round 1).  The T-th largest sampled key is then about the (stride x T)-th key
//     of the graph: with ~32 k sampled edges on C2 the certified subgraph still holds 1.47 M triangles for T = 50 k.
//   TOP = true: the `target` HEAVIEST edges (weight at or above the lower edge of the bin of a 256-bin weight histogram
//     where the count from the top reaches target; es_hist_kernel).  A top triangle has three heavy edges, so its first
//     edge is far more likely to be in this set than in a uniform sample of the same size.  A block takes a contiguous
//     chunk of the edge list, compacts the qualifying edges into LDS (coalesced read of es, ballot prefix) and deals
//     them to its groups.
constexpr int SM_CHUNK = 256;  // edges per block and chunk in the TOP form (small: a chunk inside an inlier row is all heavy edges)

// Weight -> bin, heavier = higher, LOGARITHMIC in 1 - s: the weights of good edges crowd against 1.0 (s = exp(-d^2 / 2
// sigma^2) with d << sigma), so linear bins would put tens of thousands of edges into the top one and the sample could
// not be sized.  1 - s is exact in fp32 for s >= 0.5; its exponent and top three mantissa bits give 8 bins per octave:
// 1 - s in [2^-24, 2^-3) -> bins 255 .. 80, s == 1 -> 255, lighter edges -> the low bins.  (wlo / wshift: unused.)
__device__ __forceinline__ uint32_t weight_bin(uint32_t wbits, uint32_t wlo, uint32_t wshift) {
  (void)wlo; (void)wshift;
  const uint32_t lb = __float_as_uint(1.0f - __uint_as_float(wbits)) >> 20;  // sign 0, exponent, 3 mantissa bits
  const int b = (int)(103u << 3) + 255 - (int)lb;                              // 2^-24 has exponent field 103
  return (uint32_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
}

template <int TG, bool TOP>
__global__ __launch_bounds__(256) void tri_sample_hist_kernel(const uint64_t* __restrict__ bits, int W,
                                                              const uint32_t* __restrict__ wpre,
                                                              const uint32_t* __restrict__ ebi,
                                                              const uint32_t* __restrict__ ebj,
                                                              const uint32_t* __restrict__ ei,
                                                              const uint32_t* __restrict__ ej,
                                                              const float* __restrict__ es, uint64_t E,
                                                              uint32_t stride, uint32_t part, uint32_t parts,
                                                              uint32_t klo, uint32_t shift,
                                                              uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ es_hist, uint64_t target,
                                                              uint32_t wlo, uint32_t wshift,
                                                              const uint64_t* __restrict__ E_dev) {
  // Launched before the host knew the count (E: what the arrays hold).  More edges than that: the CSR indices this kernel
  // derives from the bit rows would point beyond the edge arrays — nothing may run (the host repeats the call: sc_capi.hip).
  if (E_dev && *E_dev > E) return;
  if (E_dev) E = *E_dev;
  __shared__ uint32_t lh[PR_BINS * PR_COPIES];  // [bin][copy]
  __shared__ uint32_t l_e[TOP ? SM_CHUNK : 1];  // TOP: the qualifying edges of the current chunk
  __shared__ uint32_t s_theta, s_cnt, s_wtot[4];
  __shared__ uint64_t plds[8];
  for (int b = threadIdx.x; b < PR_BINS * PR_COPIES; b += 256) lh[b] = 0;
  if (TOP) {  // the weight bin at which the count from the top reaches `target` (no such bin: every edge qualifies)
    static_assert(PR_BINS == 256, "one bin per thread, walked from the top");
    const uint32_t bin = PR_BINS - 1 - threadIdx.x;
    uint64_t mine = 0;
    for (int c = 0; c < PR_HCOPIES; c++) mine += es_hist[c * PR_BINS + bin];
    if (threadIdx.x == 0) s_theta = 0u;
    uint64_t tot;
    const uint64_t before = block_exscan_u64(mine, plds, &tot);  // (its barriers also order the s_theta default)
    if (before < target && target <= before + mine) s_theta = bin;
  }
  __syncthreads();
  const int gl = threadIdx.x & (TG - 1);
  const uint64_t n_s = (E + stride - 1) / stride;  // sampled edges: e = g * stride, g = 0 .. n_s - 1
  // this launch takes the sampled edges g = part, part + parts, ... (one process per GPU: every rank samples its share
  // and the histograms are summed by an all-reduce — distinct edges give distinct triangles, so the sum certifies)
  const uint64_t n_loc = n_s > part ? (n_s - part + parts - 1) / parts : 0;
  const uint64_t groups = TOP ? (256 / TG) : (uint64_t)gridDim.x * (256 / TG);
  const uint32_t theta = TOP ? s_theta : 0u;
  // a chunk is SM_CHUNK x parts consecutive edges, of which this rank considers every parts-th: the same number of
  // candidates per block whatever the number of ranks sharing the sample (per-block fixed costs stay amortised)
  const uint64_t chunk_edges = (uint64_t)SM_CHUNK * parts;
  const uint64_t n_chunks = TOP ? (E + chunk_edges - 1) / chunk_edges : 1;
  for (uint64_t ch = TOP ? blockIdx.x : 0; ch < n_chunks; ch += TOP ? gridDim.x : 1) {
  uint64_t q_end = n_loc;
  if (TOP) {
    // compaction of the chunk's qualifying edges, in edge order: four rounds of 256 edges (ballot prefix per wave,
    // wave totals through LDS); part / parts: this rank takes the qualifying edges with e % parts == part
    if (threadIdx.x == 0) s_cnt = 0u;
    __syncthreads();
    for (uint32_t r = 0; r < (SM_CHUNK / 256) * parts; r++) {
      // thread t of round r looks at edge (chunk base + r * 256 + t) only if it is this rank's (e % parts == part):
      // the edges are dealt so that consecutive threads of a round see consecutive edges of THIS rank
      const uint64_t e = ch * chunk_edges + ((uint64_t)r * 256 + threadIdx.x);
      const bool ok = e < E && (e % parts) == part && weight_bin(__float_as_uint(es[e]), wlo, wshift) >= theta;
      const uint64_t bal = __ballot(ok);
      const int wave = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) s_wtot[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t pos = s_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      for (int w = 0; w < wave; w++) pos += s_wtot[w];
      if (ok) l_e[pos] = (uint32_t)e;
      __syncthreads();
      if (threadIdx.x == 0) s_cnt += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
      __syncthreads();
    }
    q_end = s_cnt;
  }
  for (uint64_t q = TOP ? (threadIdx.x / TG) : (uint64_t)blockIdx.x * (256 / TG) + (threadIdx.x / TG); q < q_end; q += groups) {
    const uint64_t e = TOP ? (uint64_t)l_e[q] : (q * parts + part) * stride;
    const uint32_t i = ei[e], j = ej[e];
    const uint32_t rowi = i * (uint32_t)W, rowj = j * (uint32_t)W;
    const int w0 = j >> 6;
    const float s_ij = es[e];
    const uint32_t bi = ebi[e], bj = ebj[e];  // the CSR bases travel with the edge: same memory level as ei / ej
    // No cross-lane step: every lane owns its words.  The loads of RB rounds are issued together, level by level (row
    // words -> prefix words -> weights).  Measured on C2: 33 us either way — the time is not the per-round latency chain
    // but gather / LDS-atomic throughput: most sampled edges join two inliers (the inlier clique holds ~60 % of all
    // edges) with ~250 common neighbours above j each, i.e. ~8 M key evaluations per call.
    constexpr int RB = 4;
    for (int wb = w0 + gl; wb < W; wb += TG * RB) {
      uint64_t ai[RB], aj[RB], m[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) {  // level 1: row words
        const int w = wb + r * TG;
        const bool live = w < W;
        ai[r] = bits[rowi + (live ? w : w0)];
        aj[r] = bits[rowj + (live ? w : w0)];
        m[r] = live ? (ai[r] & aj[r]) : 0ull;
        if (w == w0) m[r] &= mask_above(j & 63);
      }
      uint32_t pi[RB], pj[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) {  // level 2: prefix words (idle slots re-read the row's first one)
        const int w = wb + r * TG;
        pi[r] = bi + wpre[rowi + (m[r] ? w : w0)];
        pj[r] = bj + wpre[rowj + (m[r] ? w : w0)];
      }
#pragma unroll
      for (int r = 0; r < RB; r++) {  // level 3: weights of the common neighbours, four at a time
        uint64_t mr = m[r];
        while (mr) {
          int b[4];
          const int nbits = pop4(mr, b);
          float s_ik[4], s_jk[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint64_t below = (1ull << b[q]) - 1ull;
            const bool live = q < nbits;
            s_ik[q] = es[live ? pi[r] + (uint32_t)__popcll(ai[r] & below) : 0u];
            s_jk[q] = es[live ? pj[r] + (uint32_t)__popcll(aj[r] & below) : 0u];
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (q < nbits)
              atomicAdd(&lh[prune_bin(__float_as_uint((s_ij + s_ik[q]) + s_jk[q]), klo, shift) * PR_COPIES +
                            (threadIdx.x & (PR_COPIES - 1))], 1u);
        }
      }
    }
  }
  if (TOP) __syncthreads();  // l_e is rewritten by the next chunk
  }  // chunks
  __syncthreads();
  // flush into one of PR_HCOPIES global copies (by block): a thousand blocks adding into the SAME 256 words serialise
  // at the memory side (~12 ns per add and address); the copies are summed by the reader
  uint32_t* __restrict__ myh = hist + (size_t)(blockIdx.x & (PR_HCOPIES - 1)) * PR_BINS;
  for (int b = threadIdx.x; b < PR_BINS; b += 256) {
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < PR_COPIES; c++) v += lh[b * PR_COPIES + ((c + threadIdx.x) & (PR_COPIES - 1))];
    if (v) atomicAdd(&myh[b], v);
  }
}

// ------------------------------------------------------------------------------------------------
// 3c. pruning by an ESTIMATED bound (r04)
//
// The certifying samples above pay for their certificate: to exhibit T genuine triangles above the bound they evaluate
// 4 - 8 M keys at C2 (25 us, the largest launch of stage B), and the bound they can certify sits well below the T-th
// key, so the pruned graph still holds 10 T triangles.  But the bound need not be certified BEFORE it is used: it can be
// verified AFTERWARDS, for free.  Take ANY value LB, prune with smin = LB - 2 - 1e-6, enumerate, select.  If the strong
// subgraph turns out to hold >= T triangles with key >= LB, the proof of 3b applies word for word (it only needs "at least
// T triangles have key >= LB") and the selection is the full graph's top-T, ties included; if not, nothing is known and the
// call is repeated with a certifying sample.  The select's first round already counts the keys in [LB, 3.0] — the check
// is one comparison where that round is resolved (select_resolve, SelectState::want_req).
// So LB only has to be a good GUESS of a key of rank ~2 T: the lower edge of the histogram bin where
// rate x (sampled count from the top) reaches margin x T, from a uniform 1-in-rate sample of ALL the graph's triangles.
// The sample: a triangle (i, j, k), i < j < k, is found from its edge (i, j) in the 64-column word k / 64 of the row pair;
// edge e looks at the words w >= j / 64 with w = hash(e) (mod rate) only — every (edge, word) pair, hence every triangle,
// with probability exactly 1 / rate, independently across edges and words, so the sample is not dominated by a few edges
// (an inlier-inlier edge has hundreds of triangles; a sampled word holds a handful).  One lane per edge: an edge record,
// two gathered row words and — where they meet — the two prefix words and two weights per triangle: 0.36 M key
// evaluations at C2 instead of 4 M, no block-level step at all.
// With margin = 2 the estimate fails when the sample overstates the density of the top keys twofold: at >= 1000 expected
// samples above the bound that is > 10 sigma even with word-sized clusters.  Failing costs a repeated call, never a wrong
// result; a context that has seen one failure stops estimating (sc_capi.hip).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t edge_hash(uint32_t e) {
  e ^= e >> 16; e *= 0x7FEB352Du; e ^= e >> 15; e *= 0x846CA68Bu; e ^= e >> 16;  // (lowbias32)
  return e;
}

constexpr uint32_t SAMPLE_CAND_BLOCKS = 256;  // workgroups of the estimating sample that leave a candidate triangle behind
__global__ __launch_bounds__(256) void tri_sample_words_kernel(const uint64_t* __restrict__ bits, int W,
                                                               const uint32_t* __restrict__ wpre,
                                                               const uint32_t* __restrict__ ebi,
                                                               const uint32_t* __restrict__ ebj,
                                                               const uint32_t* __restrict__ ei,
                                                               const uint32_t* __restrict__ ej,
                                                               const float* __restrict__ es, uint64_t E, uint32_t rmask,
                                                               uint32_t* __restrict__ hist,
                                                               const uint64_t* __restrict__ E_dev,
                                                               const uint32_t* __restrict__ ebase, uint4* __restrict__ cand,
                                                               unsigned long long* __restrict__ cand_slot) {
  // ebase (with ebi == ebj == nullptr): the per-row CSR bases are looked up (after launch_edge_build)
  // cand (optional): the best-keyed triangle this workgroup sampled, {key bits, i, j, k} (key 0: none) — the voters of stage
  // C2's reference frame (sc_gramref.hpp)
  if (E_dev && *E_dev > E) return;  // (launched before the host knew the count: see tri_sample_hist_kernel)
  if (E_dev) E = *E_dev;
  // a host-free call's grid covers more edges than there are: the workgroups beyond them leave before they clear 16 KiB of LDS
  // (one that would have left a candidate says "none": cand is not cleared between calls)
  if ((uint64_t)blockIdx.x * 256 >= E) {
    if (cand != nullptr && blockIdx.x < SAMPLE_CAND_BLOCKS && threadIdx.x == 0) cand[blockIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  uint32_t best_key = 0u, best_e = 0u, best_k = 0u;  // (E < 2^32)
  // only the first SAMPLE_CAND_BLOCKS workgroups keep their best triangle (tracking it in every one cost the launch 3.6 us at C2;
  // 64 voters want 64 good triangles, and the best of these workgroups' ~50 000 sampled keys per voter is that)
  const bool track = cand != nullptr && blockIdx.x < SAMPLE_CAND_BLOCKS;
  __shared__ uint32_t lh[PR_BINS * EST_COPIES];
  for (int b = threadIdx.x; b < PR_BINS * EST_COPIES; b += 256) lh[b] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += stride) {
    const uint32_t j = ej[e], i = ei[e];
    const int w0 = (int)(j >> 6);
    // the first word >= w0 in this edge's residue class (hashed by its END POINTS: the sample does not depend on the edge numbering)
    int w = w0 + (int)((edge_hash(i * 0x9E3779B1u + j) - (uint32_t)w0) & rmask);
    if (w >= W) continue;
    const uint32_t rowi = i * (uint32_t)W, rowj = j * (uint32_t)W;
    for (; w < W; w += (int)rmask + 1) {
      const uint64_t ai = bits[rowi + w], aj = bits[rowj + w];
      uint64_t m = ai & aj;
      if (w == w0) m &= mask_above((int)(j & 63));
      if (m == 0) continue;
      // only the lanes that found a triangle pay for the edge's weight and CSR bases (one memory level, beside the prefix words)
      const float s_ij = es[e];
      const uint32_t bi = ebi ? ebi[e] : ebase[i], bj = ebj ? ebj[e] : ebase[j];
      const uint32_t pi = bi + wpre[rowi + w], pj = bj + wpre[rowj + w];
      while (m) {
        int b[4];
        const int nbits = pop4(m, b);
        float s_ik[4], s_jk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint64_t below = (1ull << b[q]) - 1ull;
          const bool live = q < nbits;
          s_ik[q] = es[live ? pi + (uint32_t)__popcll(ai & below) : 0u];
          s_jk[q] = es[live ? pj + (uint32_t)__popcll(aj & below) : 0u];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q < nbits) {
            const uint32_t kb = __float_as_uint((s_ij + s_ik[q]) + s_jk[q]);  // (keys are positive floats: their bits order like they do)
            atomicAdd(&lh[est_bin(kb) * EST_COPIES + (threadIdx.x & (EST_COPIES - 1))], 1u);
            if (track && kb > best_key) { best_key = kb; best_e = (uint32_t)e; best_k = (uint32_t)(64 * w + b[q]); }
          }
      }
    }
  }
  __syncthreads();
  uint32_t* __restrict__ myh = hist + (size_t)(blockIdx.x & (PR_HCOPIES - 1)) * PR_BINS;
  for (int b = threadIdx.x; b < PR_BINS; b += 256) {
    uint32_t v = 0;
#pragma unroll
    for (int c = 0; c < EST_COPIES; c++) v += lh[b * EST_COPIES + ((c + threadIdx.x) & (EST_COPIES - 1))];
    if (v) atomicAdd(&myh[b], v);
  }
  if (track) {  // (workgroup-uniform) the workgroup's best: by key, then by lowest thread — the sample is deterministic, so is this
    __shared__ uint32_t s_best[4], s_who[4];
    uint32_t bk = best_key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bk = max(bk, (uint32_t)__shfl_xor((int)bk, o));
    const uint64_t holders = __ballot(best_key == bk);
    if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = bk; s_who[threadIdx.x >> 6] = (threadIdx.x & ~63u) + (uint32_t)(__ffsll((unsigned long long)holders) - 1); }
    __syncthreads();
    uint32_t wk = s_best[0], wt = s_who[0];
#pragma unroll
    for (int v = 1; v < 4; v++)
      if (s_best[v] > wk) { wk = s_best[v]; wt = s_who[v]; }  // (waves in order: equal keys keep the lower thread)
    if (threadIdx.x == wt) {
      cand[blockIdx.x] = wk ? make_uint4(best_key, ei[best_e], ej[best_e], best_k) : make_uint4(0u, 0u, 0u, 0u);
      // voter v of the frame = the best candidate among the workgroups = v (mod 64): one 64-bit max per workgroup (ControlBlock::ref_slot, zeroed per call)
      if (wk) atomicMax(&cand_slot[blockIdx.x & 63u], ((unsigned long long)wk << 32) | (unsigned long long)blockIdx.x);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 1'. edge_build (r04): everything between stage A and the estimating sample in ONE launch — the hot path's form of
//     row_stats_scan + edge_fill (two launches, 29.5 us at C2).
//
// What made the row kernel a launch of its own was the prefix over the rows: a wave cannot place row i's edges before it
// knows how many edges the rows before it hold, and counting them meant reading their bit rows.  Stage A now leaves
// deg+[i] behind (atomics where the adjacency words are made, sc_compat.hip), so a workgroup of EBK_ROWS waves sums
// deg+[0 .. its first row) itself — a few 16-byte loads per thread, no look-back, no cross-workgroup dependency — and then
// each wave, for its row: word-prefix popcounts (wpre), the CSR offset and base, the strong-bit row cleared, the row's
// edges with their recomputed weights (as edge_fill_kernel).
// (r04 also took the ESTIMATING sample here — sampled triangles recomputed from the correspondences through a per-wave LDS
// queue: 50.3 us against 16.4 + 13.7 for the two launches, a dense row being one wave's serial chain; removed in r05,
// profiles/r04_ab_edge_build.txt keeps the numbers.)
// The CSR bases of an edge's OTHER end (ebj) are not known here (row j's wave may not have run): the counting pass looks
// them up in ebase instead (one more gather, beside its row loads).  n <= 64 * EBK_WMAX only.
// ------------------------------------------------------------------------------------------------
constexpr int EBK_ROWS = 8;     // rows (waves) per workgroup
constexpr int EBK_CH = 256;     // column indices staged per wave and chunk
constexpr int EBK_WMAX = 320;   // words per bit row it can handle (n <= 20 480): NCH = 2 chunks of 64 words up to 8192, 5 beyond

template <int NCH>  // 64-word chunks of a bit row: W <= 64 NCH
__global__ __launch_bounds__(64 * EBK_ROWS) void edge_build_kernel(const uint64_t* __restrict__ bits,
                                                                   const float* __restrict__ planes, Derived dv, int n,
                                                                   int ld, int W, const uint32_t* __restrict__ degp,
                                                                   uint32_t* __restrict__ wpre,
                                                                   uint64_t* __restrict__ zero_rows,
                                                                   uint64_t* __restrict__ edge_off,
                                                                   uint32_t* __restrict__ ebase,
                                                                   uint32_t* __restrict__ ei, uint32_t* __restrict__ ej,
                                                                   float* __restrict__ es, uint64_t cap,
                                                                   uint64_t* __restrict__ host_total,
                                                                   uint64_t* __restrict__ live_range) {
  __shared__ uint64_t s_red[EBK_ROWS];
  __shared__ uint32_t l_j[EBK_ROWS][EBK_CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * EBK_ROWS, i = r0 + wave;
  // edges of the rows before this workgroup's (r0 is a multiple of 4: whole 16-byte pieces)
  uint64_t acc = 0;
  {  // (four 16-byte loads in flight: the loop's trips were dependent round trips — up to three for the last rows at C2, ten at C3)
    constexpr int QS = 64 * EBK_ROWS * 4;
    int q = tid * 4;
    for (; q + 3 * QS < r0; q += 4 * QS) {
      const uint4 v0 = *reinterpret_cast<const uint4*>(degp + q), v1 = *reinterpret_cast<const uint4*>(degp + q + QS),
                  v2 = *reinterpret_cast<const uint4*>(degp + q + 2 * QS), v3 = *reinterpret_cast<const uint4*>(degp + q + 3 * QS);
      acc += ((uint64_t)v0.x + v0.y + v0.z + v0.w) + ((uint64_t)v1.x + v1.y + v1.z + v1.w) + ((uint64_t)v2.x + v2.y + v2.z + v2.w) +
             ((uint64_t)v3.x + v3.y + v3.z + v3.w);
    }
    uint4 w[3];
    int nw = 0;
#pragma unroll
    for (int u = 0; u < 3; u++) { const bool in = q + u * QS < r0; w[u] = in ? *reinterpret_cast<const uint4*>(degp + q + u * QS) : make_uint4(0u, 0u, 0u, 0u); nw += in; }
#pragma unroll
    for (int u = 0; u < 3; u++) acc += (uint64_t)w[u].x + w[u].y + w[u].z + w[u].w;
    (void)nw;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) s_red[wave] = acc;
  __syncthreads();
  if (i >= n) return;
  uint64_t my_off = 0;
#pragma unroll
  for (int w8 = 0; w8 < EBK_ROWS; w8++) my_off += s_red[w8];
  for (int r = r0; r < i; r++) my_off += degp[r];  // wave-uniform: scalar loads
  const int wi = i >> 6, bi = i & 63;
  const float4* __restrict__ aos4 = reinterpret_cast<const float4*>(planes + 6 * (size_t)ld);
  const float pix = planes[i], piy = planes[ld + i], piz = planes[2 * ld + i];
  const float qix = planes[3 * ld + i], qiy = planes[4 * ld + i], qiz = planes[5 * ld + i];
  // ---- the row's words: prefix popcounts, the strong row cleared
  uint32_t d_all = 0, d_low = 0;
  uint64_t up[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int w = 64 * c + lane;
    const uint64_t v = w < W ? bits[(size_t)i * W + w] : 0ull;
    const uint32_t pc = (uint32_t)__popcll(v);
    uint32_t tot;
    const uint32_t ex = group_exscan<64>(pc, &tot);
    if (w < W) {
      wpre[(size_t)i * W + w] = d_all + ex;
      zero_rows[(size_t)i * W + w] = 0ull;
    }
    d_all += tot;
    const uint64_t below = (1ull << bi) - 1ull;
    d_low += w < wi ? pc : (w == wi ? (uint32_t)__popcll(v & below) : 0u);
    up[c] = w < wi ? 0ull : (w == wi ? (v & mask_above(bi)) : v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d_low += __shfl_xor(d_low, o);
  const uint32_t my_base = (uint32_t)my_off - d_low;
  if (lane == 0) { edge_off[i] = my_off; ebase[i] = my_base; }
  // ---- edges, chunk by chunk
  uint64_t ebase_row = my_off;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int w = 64 * c + lane;
    const uint64_t v = up[c];
    uint32_t tot;
    const uint32_t r = group_exscan<64>((uint32_t)__popcll(v), &tot);
    for (uint32_t c0 = 0; c0 < tot; c0 += EBK_CH) {  // wave-uniform
      uint64_t vv = v;
      uint32_t rr = r - c0;  // modular: positions outside [0, EBK_CH) are skipped
      while (vv) {
        const int b = __builtin_ctzll(vv);
        vv &= vv - 1;
        if (rr < (uint32_t)EBK_CH) l_j[wave][rr] = (uint32_t)(w * 64 + b);
        rr++;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS is in-order within a wave
      const uint32_t cnt = min((uint32_t)EBK_CH, tot - c0);
      for (uint32_t t = lane; t < cnt; t += 64) {
        const uint32_t j = l_j[wave][t];
        const uint64_t e = ebase_row + c0 + t;
        if (e >= cap) continue;
        const float4 a4 = aos4[2 * (size_t)j], b4 = aos4[2 * (size_t)j + 1];
        bool edge;
        const float sw = pair_weight(dist3(pix, piy, piz, a4.x, a4.y, a4.z), dist3(qix, qiy, qiz, a4.w, b4.x, b4.y), dv.d_thr, dv.min_len,
                                     dv.neg_inv2sig2, edge);
        ei[e] = (uint32_t)i; ej[e] = j; es[e] = sw;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // l_j is rewritten by the next chunk
    }
    ebase_row += tot;
  }
  if (i == n - 1 && lane == 0) {  // the last row's wave knows the edge count: the host polls for it
    edge_off[n] = ebase_row;
    if (live_range) { live_range[0] = 0ull; live_range[1] = ebase_row; }
    if (host_total) publish_host(host_total, ebase_row);  // (nullptr: a host-free call reads the count from edge_off[n] — DeferredPub)
  }
}

bool edge_build_fits(int n) { return n <= 64 * EBK_WMAX; }

void launch_edge_build(const Graph& g, const Points& pts, const Derived& dv, uint64_t* zero_rows, uint64_t* edge_off, uint32_t* ebase,
                       uint32_t* ei, uint32_t* ej, float* es, uint64_t cap, uint64_t* host_total, hipStream_t st, uint64_t* live_range) {
  const dim3 grid((g.n + EBK_ROWS - 1) / EBK_ROWS), block(64 * EBK_ROWS);
  if (g.W <= 128)
    hipLaunchKernelGGL(edge_build_kernel<2>, grid, block, 0, st, g.bits, pts.planes, dv, g.n, g.ld, g.W, g.degp,
                       const_cast<uint32_t*>(g.wpre), zero_rows, edge_off, ebase, ei, ej, es, cap, host_total, live_range);
  else
    hipLaunchKernelGGL(edge_build_kernel<5>, grid, block, 0, st, g.bits, pts.planes, dv, g.n, g.ld, g.W, g.degp,
                       const_cast<uint32_t*>(g.wpre), zero_rows, edge_off, ebase, ei, ej, es, cap, host_total, live_range);
}

// sum of the copies -> one 256-bin histogram (the form the ranks exchange)
__global__ __launch_bounds__(256) void hist_reduce_kernel(const uint32_t* __restrict__ copies, uint32_t* __restrict__ out) {
  uint32_t v = 0;
#pragma unroll
  for (int c = 0; c < PR_HCOPIES; c++) v += copies[c * PR_BINS + threadIdx.x];
  out[threadIdx.x] = v;
}
void launch_hist_reduce(const uint32_t* copies, uint32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(hist_reduce_kernel, dim3(1), dim3(256), 0, st, copies, out);
}

// every block derives smin from the histogram (256 bins: cheap) and sets the strong bits of its edges in `mbits`
// (zeroed beforehand; only upper-triangle entries are needed: bit j of row i for i < j).  Block 0 publishes smin.
__global__ __launch_bounds__(256) void prune_bits_kernel(const uint32_t* __restrict__ hist, int copies, uint64_t want,
                                                         uint32_t klo, uint32_t shift,
                                                         const uint32_t* __restrict__ ei,
                                                         const uint32_t* __restrict__ ej,
                                                         const float* __restrict__ es, uint64_t E, int W,
                                                         unsigned long long* __restrict__ mbits,
                                                         float* __restrict__ smin_out,
                                                         uint32_t* __restrict__ klb_out, StrongList sl,
                                                         uint32_t* __restrict__ tcnt,
                                                         const uint64_t* __restrict__ own,
                                                         const uint64_t* __restrict__ E_dev, int trimmed) {
  // E: what the grid and the arrays cover; Ea: the edges there really are (E_dev: launched before the host knew).  More than
  // the arrays hold: nothing may run — no strong edge is listed, so the counting and key kernels that follow find no work
  // (the host repeats the call)
  if (E_dev && *E_dev > E) return;
  const uint64_t Ea = E_dev ? *E_dev : E;
  // a host-free call's grid covers more edges than there are: the workgroups beyond them have nothing to list, and their tcnt
  // entries are never read (the scan of a host-free call skips the tiles beyond ControlBlock::live_edges and treats what lies
  // beyond the count as zero) — they leave before the histogram walk
  if (trimmed && E_dev && sl.region_blocks == 0 && (uint64_t)blockIdx.x * 256 >= Ea) return;
  __shared__ uint64_t lds[8];
  __shared__ float s_smin;
  __shared__ uint32_t s_klb, s_base, s_wcnt[4];
  static_assert(PR_BINS == 256, "one bin per thread, walked from the top");
  const uint32_t bin = PR_BINS - 1 - threadIdx.x;
  uint64_t mine = 0;
  if (copies == PR_HCOPIES) {  // (the sample's own copies: four loads in flight — a loop of `copies` trips is four round trips to hipcc)
    static_assert(PR_HCOPIES == 4, "four copies");
    const uint32_t h0 = hist[bin], h1 = hist[PR_BINS + bin], h2 = hist[2 * PR_BINS + bin], h3 = hist[3 * PR_BINS + bin];
    mine = (uint64_t)h0 + h1 + h2 + h3;
  } else {
    for (int c = 0; c < copies; c++) mine += hist[c * PR_BINS + bin];  // (1: summed by the caller)
  }
  if (threadIdx.x == 0) { s_smin = -1.0f; s_klb = 0u; }  // default: no certified bound -> every edge is strong
  uint64_t tot;
  const uint64_t before = block_exscan_u64(mine, lds, &tot);
  if (before < want && want <= before + mine && bin > 0) {
    const float lb = shift == PR_LOGBINS ? est_bin_floor(bin) : __uint_as_float(klo + (bin << shift));  // lower edge of the crossing bin
    s_smin = (lb - 2.0f) - 1e-6f;
    s_klb = __float_as_uint(lb);  // >= want triangles have a key >= this one: the select may ignore anything below
  }
  __syncthreads();
  const float smin = s_smin;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *smin_out = smin; *klb_out = s_klb; }
  // region = blockIdx % ST_SHARDS in both forms (blocks that run side by side add to different counters: with consecutive
  // blocks on one counter this kernel took 61 us instead of 33 at C3); with contiguous regions the block's 256 EDGES are what
  // moves: block b takes chunk (b % ST_SHARDS) * region_blocks + b / ST_SHARDS
  const uint32_t region = blockIdx.x & (ST_SHARDS - 1);
  const uint64_t chunk = sl.region_blocks ? (uint64_t)region * sl.region_blocks + blockIdx.x / ST_SHARDS : (uint64_t)blockIdx.x;
  const uint64_t e = (sl.region_blocks && blockIdx.x / ST_SHARDS >= sl.region_blocks) ? E : chunk * 256 + threadIdx.x;
  bool strong = e < Ea && es[e] >= smin;
  if (strong) {  // the strong bit matrix is whole on every rank: membership of ANY vertex pair is looked up in it
    const uint32_t i = ei[e], j = ej[e];
    atomicOr(&mbits[(size_t)i * W + (j >> 6)], 1ull << (j & 63));
  }
  // sharded stage B: only this rank's edge range [own[0], own[1]) enters the list of edges to enumerate
  if (own) strong = strong && e >= own[0] && e < own[1];
  if (sl.list) {
    // the strong edges, compacted (any order: everything downstream is indexed by the edge id): one atomic per block
    // on one of ST_SHARDS counters; region r receives the blocks with blockIdx % ST_SHARDS == r (or, StrongList::region_blocks,
    // a run of consecutive blocks: ceil(blocks / ST_SHARDS) of them), so sl.cap = ceil(blocks / ST_SHARDS) * 256 entries can
    // never overflow.  Weak edges carry no triangle: count 0.
    if (e < E && (!strong || sl.region_blocks)) tcnt[e] = 0u;
    const uint64_t bal = __ballot(strong);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_wcnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {  // ONE atomic per block (returning atomics on a shared address cost ~170 ns each here)
      const uint32_t tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
      s_base = tot ? atomicAdd(&sl.fill[region], tot) : 0u;
    }
    __syncthreads();
    if (strong) {
      uint32_t pos = s_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      for (int w = 0; w < wave; w++) pos += s_wcnt[w];
      sl.list[(uint64_t)region * sl.cap + pos] = (uint32_t)e;
    }
  }
}

uint32_t strong_list_cap(uint64_t E) {
  const uint64_t blocks = (E + 255) / 256;
  return (uint32_t)(((blocks + ST_SHARDS - 1) / ST_SHARDS) * 256);
}
size_t strong_list_bytes(uint64_t E) { return (size_t)strong_list_cap(E) * ST_SHARDS * sizeof(uint32_t); }


// histogram window in key space: [bits(key_floor), bits(3.0f)], monotone in the value; (khi - klo) >> shift < 256
static void prune_window(float key_floor, uint32_t* klo_out, uint32_t* shift_out) {
  const uint32_t khi = 0x40400000u;  // 3.0f
  uint32_t klo;
  memcpy(&klo, &key_floor, 4);
  if (!(key_floor > 0.f) || klo >= khi) klo = 0x3F800000u;  // 1.0f
  const uint32_t range = khi - klo;
  const int bitsn = 32 - __builtin_clz(range);
  *klo_out = klo;
  *shift_out = bitsn > 8 ? (uint32_t)(bitsn - 8) : 0u;
}

void launch_sample_hist(const Graph& g, const uint32_t* ebi, const uint32_t* ebj, const uint32_t* ei,
                        const uint32_t* ej, const float* es, uint64_t E, uint64_t want, float key_floor, uint32_t part,
                        uint32_t parts, uint32_t* hist, uint32_t* es_hist, const Tuning& tn, hipStream_t st,
                        const uint64_t* E_dev, uint64_t E_hint) {
  uint32_t klo, shift;
  prune_window(key_floor, &klo, &shift);
  const uint64_t Ed = E_hint ? E_hint : E;  // the count the host-side choices are made with
  // Which form (r02 sweeps, profiles/r02_ab_heaviest_edge_sample.txt and r02_ab_sample_size.txt): the heaviest edges
  // certify a much tighter bound per sampled edge, but each of them is a costly one (an edge between two inliers has
  // hundreds of common neighbours).  That pays when T is a small part of the graph's triangles — C3 (N = 20 000, T = 200 k):
  // stage B 962 -> 551 us, 10.6 M -> 2.1 M enumerated — is a wash on C2 (167 vs 165 us) and loses where T is most of
  // what there is (C4, T = 500 k of 8.8 M triangles: the uniform sample at stride 1 already certifies 2 T: 211 vs 248 us).
  // C1 (N = 2000, T = 10 k) loses too: 94 vs 82 us; C2, a wash with every stage bracketed, gains on the hot path (stage B
  // 154.5 -> 138.5 us: fewer keys also means the speculative launches finish sooner).  What separates them is how small T
  // is against the triangles the graph holds, for which E^2 / N is a proxy known before any is counted: C3 2550 T, C2
  // 890 T, C1 320 T, C4 90 T.
  // sample_mode: 0 = by this rule (E^2 / N >= 600 T), 1 = every stride-th edge, 2 = the heaviest edges.
  const bool top = tn.sample_mode == 2 ||
                   (tn.sample_mode == 0 && (double)Ed * (double)Ed / (double)(g.n > 0 ? g.n : 1) >= 600.0 * (double)want);
  if (top && es_hist) {
    // the heaviest edges (TOP form): weight histogram, then the sample itself
    uint32_t wlo, wshift;
    {
      const float wfloor = key_floor / 3.0f;
      uint32_t lo; memcpy(&lo, &wfloor, 4);
      const uint32_t hi = 0x3F800000u;  // 1.0f
      if (!(wfloor > 0.f) || lo >= hi) lo = 0x3F000000u;  // 0.5f
      const int bitsn = 32 - __builtin_clz(hi - lo);
      wlo = lo; wshift = bitsn > 8 ? (uint32_t)(bitsn - 8) : 0u;
    }
    uint64_t target = tn.sample_edges ? tn.sample_edges : (want / 3 < 16384 ? 16384 : want / 3);
    const int tg = tn.tg_sample ? tn.tg_sample : (g.W > 256 ? 32 : 16);  // C3 (W = 313): 134 (16) vs 107 us (32); C2: 24.4 vs 25.9
    uint64_t nb = (E + (uint64_t)SM_CHUNK * parts - 1) / ((uint64_t)SM_CHUNK * parts);
    if (nb > 8192) nb = 8192;
    if (tn.sample_blocks) nb = tn.sample_blocks;
#define SC_LAUNCH_TOP(TGV) hipLaunchKernelGGL((tri_sample_hist_kernel<TGV, true>), dim3((unsigned)nb), dim3(256), 0, st, g.bits, g.W, g.wpre, ebi, ebj, ei, ej, es, E, 1u, part, parts, klo, shift, hist, es_hist, target, wlo, wshift, E_dev)
    if (tg == 4) SC_LAUNCH_TOP(4); else if (tg == 8) SC_LAUNCH_TOP(8); else if (tg == 32) SC_LAUNCH_TOP(32); else if (tg == 64) SC_LAUNCH_TOP(64); else SC_LAUNCH_TOP(16);
#undef SC_LAUNCH_TOP
    return;
  }
  // every R-th edge.  The sample must grow with T: the bound is the T-th largest SAMPLED key, so a small sample of a
  // large T certifies little.  ~5T/8 sampled edges (>= 32k) is the sweet spot on C2 for T = 50k ... 400k (swept again
  // at the end of round 1); Tuning::sample_edges overrides.
  uint64_t target = want * 5 / 8;
  if (target < 32768) target = 32768;
  if (tn.sample_edges) target = tn.sample_edges;
  uint64_t stride = Ed / (target ? target : 1);
  if (stride < 1) stride = 1;
  if (stride > 64) stride = 64;
  const uint64_t n_s = (E + stride - 1) / stride;
  const uint64_t n_loc = n_s > part ? (n_s - part + parts - 1) / parts : 0;
  if (n_loc == 0) return;
  // lanes per sampled edge (r02 sweep, profiles/r02_ab_lanes_per_edge.txt): 16 while the sample is small enough for
  // about one edge per group (C2 26.9 -> 19.8 us, C3 129 -> 112), 8 once groups take several trips (C4: 74 vs 81)
  const int tg = tn.tg_sample ? tn.tg_sample : ((n_loc <= 131072 || g.W > 256) ? 16 : 8);
  uint64_t nb = (n_loc + (256 / tg) - 1) / (256 / tg);
  // about one sampled edge per group (measured: 256 blocks 70 us, 1024+ blocks 35 us on C2)
  if (nb > 4096) nb = 4096;
#define SC_LAUNCH_SAMPLE(TGV) hipLaunchKernelGGL((tri_sample_hist_kernel<TGV, false>), dim3((unsigned)nb), dim3(256), 0, st, g.bits, g.W, g.wpre, ebi, ebj, ei, ej, es, E, (uint32_t)stride, part, parts, klo, shift, hist, (const uint32_t*)nullptr, (uint64_t)0, 0u, 0u, E_dev)
  if (tg == 4) SC_LAUNCH_SAMPLE(4); else if (tg == 8) SC_LAUNCH_SAMPLE(8); else if (tg == 32) SC_LAUNCH_SAMPLE(32); else if (tg == 64) SC_LAUNCH_SAMPLE(64); else SC_LAUNCH_SAMPLE(16);
#undef SC_LAUNCH_SAMPLE
}

SamplePlan sample_plan(uint64_t want, bool allow_estimate, const Tuning& tn, uint64_t E, int W) {
  SamplePlan sp{false, 1u, want};
  if (!allow_estimate || tn.no_estimate || tn.sample_mode != 0 || want == 0) return sp;
  // rate: 64 while 2 T still leaves >= 1024 expected samples above the bound, halved below that, never under 8 (a whole-graph
  // enumeration is what the pruning is there to avoid)
  uint32_t rate = 64;
  while (rate > 8 && 2 * want / rate < 1024) rate >>= 1;
  // ... and beyond 64 on big graphs: the sample looks at E x (row words) / rate word pairs — 16 M random 8-byte gathers at C3
  // (N = 20 000, rate 64: 156 us) for 6000 samples above the bound where 1500 do (rate 256: 45 us)
  while (rate < 512 && 2 * want / (2 * rate) >= 1024 && (double)E * (double)(W > 0 ? W : 1) / 2.0 / rate > 2.0e6) rate <<= 1;  // (an edge has ~W / 2 words beyond its higher end)
  // the rank the bound aims at, in % of T: T plus FIVE standard deviations of the sampled count at rank T — a sampled word
  // brings its triangles together, ~8 at a time — within [115, 200] (C2 151, C3 151 at its rate of 256, C4 116).  Eight until r05
  // (C2 181): over 30 000 frames of 256 distinct C2 scenes no estimate failed at 130 % nor at 115 %, 17 did at 105 % — and the bound is
  // the lower edge of the histogram bin that holds the aimed rank (256 log bins: ~0.4 T keys per bin up there), which adds to
  // the margin more often than not.  A failed estimate costs a repeated call, never a result.  C2: 186 k -> 151 k triangles
  // enumerated, - 1.9 us per frame.
  uint64_t margin = tn.est_margin_pct;
  if (!margin) {
    const double at_T = (double)want / rate;
    double m = 1.0 + 5.0 * __builtin_sqrt(8.0 / (at_T > 1.0 ? at_T : 1.0));
    m = m < 1.15 ? 1.15 : (m > 2.0 ? 2.0 : m);
    margin = (uint64_t)(m * 100.0 + 0.5);
  }
  const uint64_t aim = (want * margin + 99) / 100;
  uint64_t hw = (aim + rate - 1) / rate;
  if (hw < 128 && !tn.est_margin_pct) hw = 128;
  if (hw < 1) hw = 1;
  sp.estimate = true; sp.rate = rate; sp.hist_want = hw;
  return sp;
}

uint32_t sample_estimate_blocks(uint64_t E, const Tuning& tn) {
  uint64_t nb = (E + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (tn.sample_blocks) nb = tn.sample_blocks;
  return (uint32_t)nb;
}
uint32_t sample_candidate_blocks(uint64_t E, const Tuning& tn) {  // how many of them write a candidate (launch_sample_estimate, cand)
  const uint32_t nb = sample_estimate_blocks(E, tn);
  return nb < SAMPLE_CAND_BLOCKS ? nb : SAMPLE_CAND_BLOCKS;
}

void launch_sample_estimate(const Graph& g, const uint32_t* ebi, const uint32_t* ebj, const uint32_t* ei, const uint32_t* ej,
                            const float* es, uint64_t E, float key_floor, uint32_t rate, uint32_t* hist, const Tuning& tn,
                            hipStream_t st, const uint64_t* E_dev, const uint32_t* ebase, uint4* cand, unsigned long long* cand_slot) {
  if (E == 0) return;
  (void)key_floor;  // (the logarithmic bins need no window)
  const uint32_t nb = sample_estimate_blocks(E, tn);
  hipLaunchKernelGGL(tri_sample_words_kernel, dim3(nb), dim3(256), 0, st, g.bits, g.W, g.wpre, ebi, ebj, ei, ej, es, E,
                     rate - 1u, hist, E_dev, ebase, cand_slot ? cand : nullptr, cand_slot);
}

void launch_prune_bits(const Graph& g, const uint32_t* hist, bool hist_is_copies, const uint32_t* ei, const uint32_t* ej, const float* es,
                       uint64_t E, uint64_t want, float key_floor, uint64_t* mbits, float* smin, uint32_t* klb,
                       const StrongList& sl, uint32_t* tcnt, const uint64_t* own, hipStream_t st, const uint64_t* E_dev, bool logbins,
                       bool trimmed) {
  uint32_t klo, shift;
  prune_window(key_floor, &klo, &shift);
  if (logbins) shift = PR_LOGBINS;  // the histogram of launch_sample_estimate
  const uint64_t blocks = sl.region_blocks ? (uint64_t)sl.region_blocks * ST_SHARDS : (E + 255) / 256;
  hipLaunchKernelGGL(prune_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, hist,
                     hist_is_copies ? PR_HCOPIES : 1, want, klo, shift, ei,
                     ej, es, E, g.W, reinterpret_cast<unsigned long long*>(mbits), smin, klb, sl, tcnt, own, E_dev, trimmed ? 1 : 0);
}

// Sharded stage B, after the certificate: what enumerating row i of the PRUNED graph costs — 32 x the triangles of a
// SAMPLE of its strong edges (every edge (i, j) with (i + j) % 32 == 0: AND + popcount of the two strong rows above j,
// exactly what the counting pass does, for one edge in 32) plus one per strong edge (the counting pass spends
// about a triangle's worth of time on an edge without any).  One wave per row: the sampled edges are taken one after the
// other (wave-uniform), the 64 lanes AND the row words in parallel.  Saturated to u32.  The prefix of these costs cuts the
// rows into the ranks' ranges: a correspondence list in keypoint order puts the inliers — and nearly all the triangles of
// the pruned graph — into a few rows, which the a-priori estimate of row_stats_kernel (every edge of the FULL graph weighs
// the same) cannot see.  (A proxy from the strong degrees alone left 1.4 x between the ranks on such a scene: outlier rows
// have strong edges but hardly a triangle.)
__global__ __launch_bounds__(256) void strong_rowcost_kernel(const uint64_t* __restrict__ mbits, int n, int W,
                                                             uint32_t* __restrict__ rowcost) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint64_t* __restrict__ ri = mbits + (size_t)i * W;
  const uint64_t pattern = 0x0000000100000001ull << ((32 - (i & 31)) & 31);  // bits b of any word with (i + 64 w + b) % 32 == 0
  uint64_t tri = 0, sdeg = 0;
  for (int wb = i >> 6; wb < W; wb += 64) {
    const int w = wb + lane;
    const uint64_t v = w < W ? ri[w] : 0ull;  // (only bits above i are ever set: the strong matrix is upper-triangular)
    sdeg += (uint64_t)__popcll(v);
    const uint64_t sm = v & pattern;
    uint64_t bal = __ballot(sm != 0);
    while (bal) {  // wave-uniform: one lane's word after the other
      const int L = __builtin_ctzll(bal);
      bal &= bal - 1;
      const int wL = wb + L;
      uint64_t mL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sm >> 32), L) << 32) |
                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sm, L);
      while (mL) {
        const int bpos = __builtin_ctzll(mL);
        mL &= mL - 1;
        const uint64_t* __restrict__ rj = mbits + (size_t)(wL * 64 + bpos) * W;
        for (int w2 = wL + lane; w2 < W; w2 += 64) {
          uint64_t m = ri[w2] & rj[w2];
          if (w2 == wL) m &= mask_above(bpos);
          tri += (uint64_t)__popcll(m);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { tri += __shfl_xor(tri, o); sdeg += __shfl_xor(sdeg, o); }
  const uint64_t cost = 32ull * tri + sdeg;
  if (lane == 0) rowcost[i] = cost > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cost;
}

// prefix of the row costs AND this rank's range in one single-block launch (n values: 80 KB at N = 20 000): rows [lo, hi)
// with lo = the first row whose cost prefix reaches rank / world of the total (the rule of shard_split_kernel).
// Wave w of 16 takes the w-th sixteenth of the rows, 256 rows per step (a lane loads four consecutive rows with one 16-byte
// load: a quarter of the dependent trips the 4-byte form made — at N = 20 000 this launch took 20 us, nearly all of it load
// latency); the wave whose segment holds a boundary walks it a second time.  rowcost holds roundup(n, 4) + 1024 entries.
__device__ __forceinline__ uint4 cost4(const uint32_t* __restrict__ rowcost, int r, int r1) {  // rows r .. r + 3, zero from r1 on
  uint4 v = *reinterpret_cast<const uint4*>(rowcost + r);
  if (r + 0 >= r1) v.x = 0u;
  if (r + 1 >= r1) v.y = 0u;
  if (r + 2 >= r1) v.z = 0u;
  if (r + 3 >= r1) v.w = 0u;
  return v;
}
__global__ __launch_bounds__(1024) void cost_split_kernel(const uint32_t* __restrict__ rowcost,
                                                          const uint64_t* __restrict__ edge_off, int n, uint32_t rank,
                                                          uint32_t world, uint32_t* __restrict__ own_row,
                                                          uint64_t* __restrict__ own_edge) {
  __shared__ uint64_t s_tot[16];
  __shared__ uint32_t s_row[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = ((n + 15) / 16 + 255) / 256 * 256, r0 = min(n, wave * seg), r1 = min(n, r0 + seg);
  uint64_t mine = 0;
#pragma unroll 4
  for (int rb = r0; rb < r1; rb += 256) {
    const uint4 v = cost4(rowcost, rb + 4 * lane, r1);
    mine += (uint64_t)v.x + v.y + v.z + v.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if (lane == 0) s_tot[wave] = mine;
  if (threadIdx.x < 2) s_row[threadIdx.x] = (uint32_t)n;
  __syncthreads();
  uint64_t pre = 0, total = 0;
  for (int w = 0; w < 16; w++) { if (w < wave) pre += s_tot[w]; total += s_tot[w]; }
  for (int which = 0; which < 2; which++) {
    const uint32_t l = rank + (uint32_t)which;
    if (l == 0 || l >= world) continue;
    const uint64_t target = (uint64_t)(((unsigned __int128)total * l) / world);
    // the first row r with (sum of the costs before r) >= target lies in this wave's segment iff pre < target <= pre + mine,
    // or it is the segment's first row (pre >= target and the wave before did not reach it: handled by the min below)
    if (pre + mine < target && wave != 15) continue;       // (wave-uniform) beyond this segment
    if (pre >= target) { if (lane == 0) atomicMin(&s_row[which], (uint32_t)min(r0, n)); continue; }
    uint64_t run = pre;
    for (int rb = r0; rb < r1; rb += 256) {
      const int r = rb + 4 * lane;
      const uint4 v = cost4(rowcost, r, r1);
      const uint64_t lane_sum = (uint64_t)v.x + v.y + v.z + v.w;
      uint64_t inc = lane_sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      const uint64_t b0 = run + inc - lane_sum;              // sum of the costs before row r
      const uint64_t b1 = b0 + v.x, b2 = b1 + v.y, b3 = b2 + v.z;
      // the first of this lane's four rows whose "before" reaches the target (4: none)
      const int k = (r < r1 && b0 >= target) ? 0 : ((r + 1 < r1 && b1 >= target) ? 1 : ((r + 2 < r1 && b2 >= target) ? 2 : ((r + 3 < r1 && b3 >= target) ? 3 : 4)));
      const uint64_t hit = __ballot(k < 4);
      if (hit) {
        const int L = __builtin_ctzll(hit);  // "before" grows with the row: the lowest lane that hit holds the first such row
        const int kk = __shfl(k, L);
        if (lane == 0) atomicMin(&s_row[which], (uint32_t)(rb + 4 * L + kk));
        break;
      }
      run += __shfl(inc, 63);
    }
    // (no row of the segment reached it: the boundary is the first row of the next segment, or n — its wave reports r0)
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const uint32_t l = rank + threadIdx.x;
    const uint32_t row = l == 0 ? 0u : (l >= world ? (uint32_t)n : s_row[threadIdx.x]);
    own_row[threadIdx.x] = row;
    own_edge[threadIdx.x] = edge_off[row];
  }
}
void launch_cost_split(const uint32_t* rowcost, const uint64_t* edge_off, int n, uint32_t rank, uint32_t world,
                       uint32_t* own_row, uint64_t* own_edge, hipStream_t st) {
  hipLaunchKernelGGL(cost_split_kernel, dim3(1), dim3(1024), 0, st, rowcost, edge_off, n, rank, world, own_row, own_edge);
}
void launch_strong_rowcost(const Graph& g, const uint64_t* mbits, uint32_t* rowcost, hipStream_t st) {
  hipLaunchKernelGGL(strong_rowcost_kernel, dim3((unsigned)((g.n + 3) / 4)), dim3(256), 0, st, mbits, g.n, g.W, rowcost);
}

// ------------------------------------------------------------------------------------------------
// 4. radix select: window [lo, lo + SEL_BINS << shift), <= 3 rounds of SEL_BITS bits down to shift 0
// ------------------------------------------------------------------------------------------------
constexpr int SEL_BITS = 12;  // key bits resolved per round
constexpr int SEL_BINS = 1 << SEL_BITS;
constexpr int SEL_PER = SEL_BINS / 256;  // bins per thread of the picking block
constexpr int SEL_THREADS = 256;
constexpr int SEL_ITEMS = 16;

// current window of the select: keys in [lo, lo + 2^wbits), binned by (key - lo) >> shift into <= SEL_BINS bins
// rounds_left: launches still to come INCLUDING this one.  The window's bits are split evenly over them (at most
// SEL_BITS per round): a 16-bit window resolved in two rounds uses 256 bins per round, not 4096 — every block flushes its
// non-zero bins with global atomics, and all per-bin loops scale with the bin count.
struct SelWindow { uint32_t lo, wbits, shift, nbins; };
__device__ __forceinline__ SelWindow select_window(const SelectState* sel, const SelSnap& cur, int rounds_left) {
  SelWindow w;
  if (!cur.started) {  // first round: the window is the key range [kmin, kmax]
    w.lo = sel->kmin;
    const uint32_t range_m1 = sel->kmax - sel->kmin;
    w.wbits = range_m1 == 0 ? 0u : (uint32_t)(32 - __builtin_clz(range_m1));
  } else {
    w.lo = cur.lo;
    w.wbits = cur.wbits;
  }
  const uint32_t rl = rounds_left > 0 ? (uint32_t)rounds_left : 1u;
  uint32_t br = (w.wbits + rl - 1u) / rl;                    // this round's share of the bits
  if (br > (uint32_t)SEL_BITS) br = (uint32_t)SEL_BITS;
  if (w.wbits > br + (uint32_t)SEL_BITS * (rl - 1u)) br = (uint32_t)SEL_BITS;  // (cannot happen for wbits <= SEL_BITS * rl)
  w.shift = w.wbits > br ? w.wbits - br : 0u;
  w.nbins = 1u << (w.wbits - w.shift);                      // bins (key - lo) >> shift can reach
  return w;
}

__device__ __forceinline__ void hist_add(uint32_t* lh, bool in, uint32_t bin) {
  // wave-uniform fast path: heavy ties put a whole wave in one bin (one LDS atomic instead of a 64-way conflict)
  const uint32_t b = in ? bin : 0xFFFFFFFFu;
  const uint32_t b0 = __builtin_amdgcn_readfirstlane(b);
  const uint64_t same = __ballot(b == b0);
  const uint64_t active = __ballot(true);
  if (same == active) {
    if (b0 != 0xFFFFFFFFu && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(active))
      atomicAdd(&lh[b0], (uint32_t)__popcll(active));
  } else if (in) {
    atomicAdd(&lh[b], 1u);
  }
}

// (Tried and dropped, r01: an EXACT histogram of the keys in [certified bound, 3.0] — ~50 k distinct fp32 values on C2 —
// filled by the key kernel with one device-scope atomic per key and read by a single pick kernel, replacing both rounds:
// bit-exact, but the atomics cost the key kernel +20 us and the pick 10-50 us, against 26 us for the two rounds.)
// One select round: every block histograms its share of the keys into the window's bins (LDS, then one global add per non-zero
// bin).  Until r05 the LAST block to finish (device-scope ticket, release / acquire) walked the bins from the top, picked the bin
// holding the want-th key and narrowed the window; now the NEXT launch does that in every workgroup's prologue (select_resolve).
// four keys of the view at logical positions 4q .. 4q + 3 (entries beyond a segment's valid count read as 0, which
// no window and no threshold ever admits: real keys are positive)
// keys the view really holds (M_dev: the launch was sized before the host knew the count)
__device__ __forceinline__ uint64_t view_count(const KeyView& v) { return v.M_dev ? min(v.M, *v.M_dev) : v.M; }

template <bool SEG>
__device__ __forceinline__ uint4 view_load4(const KeyView& v, uint64_t q) {
  if (!SEG) return reinterpret_cast<const uint4*>(v.base)[q];
  const uint64_t x = q << 2, sg = x / v.seg_len, off = x - sg * v.seg_len;  // seg_len % 4 == 0: no straddling
  const uint64_t nv = v.valid[sg * v.valid_stride];
  uint4 k = make_uint4(0u, 0u, 0u, 0u);
  if (off < nv) {
    k = *reinterpret_cast<const uint4*>(v.base + sg * v.seg_stride + off);
    if (off + 1 >= nv) k.y = 0u;
    if (off + 2 >= nv) k.z = 0u;
    if (off + 3 >= nv) k.w = 0u;
  }
  return k;
}

// The state after round r from the state before it and the round's histogram.  Whole workgroup (SEL_THREADS threads, all
// arrive); lds: 10 words of 8 bytes.  Thread t owns `per` bins counted from the TOP: bins nbins - 1 - (per t + k).  The bins were
// filled by the previous launch's L2-side atomics: plain loads see them.  *shortfall: a pruning bound promised want_req keys at
// or above the window's floor and did not keep it (an estimate set too high — sc_tri.hip 3c): the selection of this call proves
// nothing; the host repeats it with a certifying sample.
__device__ SelSnap select_resolve(const SelectState* __restrict__ sel, int r, int rounds, uint64_t* lds, bool* shortfall) {
  static_assert(SEL_THREADS == 256 && SEL_PER == 16, "256 threads, at most 16 bins each");
  const SelSnap prev = sel->st[r];
  *shortfall = false;
  if (prev.done) return prev;  // (an earlier round already came down to one key value: round r added nothing)
  const SelWindow win = select_window(sel, prev, rounds - r);
  const int nbins = (int)win.nbins;
  const int per = nbins >= SEL_THREADS ? nbins / SEL_THREADS : 1;
  const bool owner = (int)threadIdx.x * per < nbins;
  const uint32_t* __restrict__ hist = sel->hist[r];
  uint32_t h[SEL_PER];
  uint64_t mine = 0;
#pragma unroll
  for (int k = 0; k < SEL_PER; k++) {
    h[k] = (owner && k < per) ? hist[nbins - 1 - ((int)threadIdx.x * per + k)] : 0u;
    mine += h[k];
  }
  const uint64_t want = prev.want, above0 = prev.above;
  if (threadIdx.x == 0) { lds[8] = 0; lds[9] = above0; }
  uint64_t tot;
  const uint64_t before = above0 + block_exscan_u64(mine, lds, &tot);
  // A window may hold fewer than want - above0 keys only where a rank selects among ITS triangles alone (sharded
  // stage B: the certified bound promises T keys above it in the whole graph, not in one rank's share): then every key
  // of the window is taken.  (Unsharded, the window always holds enough and this changes nothing.)
  const uint64_t want_eff = want < above0 + tot ? want : above0 + tot;
  *shortfall = sel->want_req != 0 && above0 + tot < sel->want_req;
  // the crossing thread: before < want <= before + mine (exactly one: the window holds >= want - above0 keys)
  if (before < want_eff && want_eff <= before + mine) {
    uint64_t run = before;
#pragma unroll
    for (int k = 0; k < SEL_PER; k++) {
      if (k < per && run < want_eff && want_eff <= run + h[k]) { lds[8] = (uint64_t)(nbins - 1 - ((int)threadIdx.x * per + k)); lds[9] = run; }
      run += h[k];
    }
  }
  __syncthreads();
  const uint32_t bin = (uint32_t)lds[8];
  const uint64_t above = lds[9];
  __syncthreads();  // (lds is the caller's again)
  SelSnap nx;
  nx.lo = win.lo + (bin << win.shift);
  nx.wbits = win.shift;  // the chosen bin is the next window
  nx.started = 1u;
  nx.done = win.shift == 0 ? 1u : 0u;
  nx.want = want_eff;
  nx.above = above;
  nx.need_eq = nx.done ? want_eff - above : 0ull;
  nx.pad = 0;
  return nx;
}

// round r of `rounds` (see SelectState): resolve round r - 1, then add this launch's share of the keys into hist[r]
template <bool SEG>
__global__ __launch_bounds__(SEL_THREADS) void select_round_kernel(KeyView view, SelectState* __restrict__ sel,
                                                                   int r, int rounds, uint64_t* __restrict__ host_short) {
  const uint64_t M = view_count(view);
  const uint32_t* __restrict__ wkey = view.base;
  __shared__ uint32_t lh[SEL_BINS];
  __shared__ uint64_t lds[10];
  for (int b = threadIdx.x; b < SEL_BINS; b += SEL_THREADS) lh[b] = 0;
  SelSnap cur;
  if (r == 0) {
    cur = sel->st[0];
  } else {
    bool shortfall;
    cur = select_resolve(sel, r - 1, rounds, lds, &shortfall);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sel->st[r] = cur;
      if (shortfall && host_short) publish_host(host_short, 1ull);
    }
  }
  if (cur.done) return;  // uniform over the grid
  const SelWindow win = select_window(sel, cur, rounds - r);
  const int nbins = (int)win.nbins;
  __syncthreads();
  const uint64_t width = 1ull << win.wbits;
  const uint64_t M4 = M >> 2;  // whole uint4 groups (wkey comes from hipMalloc: 16-byte aligned)
  const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
  auto bin4 = [&](const uint4& k4) {
    const uint32_t ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t rel = (uint64_t)ks[c] - (uint64_t)win.lo;  // wraps huge when key < lo
      hist_add(lh, (ks[c] >= win.lo) && (rel < width), (uint32_t)(rel >> win.shift));
    }
  };
  // (hipcc does not carry loads across the trips of a loop: four grid-strided loads are issued together — a thread's SEL_ITEMS keys
  // used to be four dependent round trips, ~2 us of the launch's 6)
  uint64_t q = (uint64_t)blockIdx.x * SEL_THREADS + threadIdx.x;
  for (; q + 3 * stride < M4; q += 4 * stride) {
    const uint4 a0 = view_load4<SEG>(view, q), a1 = view_load4<SEG>(view, q + stride), a2 = view_load4<SEG>(view, q + 2 * stride),
                a3 = view_load4<SEG>(view, q + 3 * stride);
    bin4(a0); bin4(a1); bin4(a2); bin4(a3);
  }
  for (; q < M4; q += stride) bin4(view_load4<SEG>(view, q));
  if (!SEG && blockIdx.x == 0 && threadIdx.x < (M & 3)) {  // tail (a segmented view is a whole number of groups)
    const uint32_t key = wkey[(M4 << 2) + threadIdx.x];
    const uint64_t rel = (uint64_t)key - (uint64_t)win.lo;
    if ((key >= win.lo) && (rel < width)) atomicAdd(&lh[(uint32_t)(rel >> win.shift)], 1u);
  }
  __syncthreads();
  uint32_t* __restrict__ hist = sel->hist[r];
  for (int b = threadIdx.x; b < nbins; b += SEL_THREADS) {
    const uint32_t v = lh[b];
    if (v) atomicAdd(&hist[b], v);
  }
}

KeyView plain_view(const uint32_t* wkey, uint64_t M) { return KeyView{wkey, M, 0, 0, nullptr, 0, nullptr}; }

void launch_select_rounds(const KeyView& view, SelectState* s, int rounds, const Tuning& tn, hipStream_t st,
                          uint64_t* host_short) {
  const uint64_t M = view.M;
  if (M == 0) return;
  uint64_t blocks = (M + (uint64_t)SEL_THREADS * SEL_ITEMS - 1) / ((uint64_t)SEL_THREADS * SEL_ITEMS);
  // 256 blocks measured best on C2 (1.5 M keys): every block pays an agent-scope release for its ticket
  uint64_t cap = 256;
  if (tn.sel_blocks >= 1) cap = tn.sel_blocks;
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
  if (rounds > 3) rounds = 3;  // (SelectState::hist; 3 x SEL_BITS covers a 32-bit key)
  for (int round = 0; round < rounds; round++) {
    if (view.seg_len) hipLaunchKernelGGL(select_round_kernel<true>, dim3((unsigned)blocks), dim3(SEL_THREADS), 0, st, view, s, round, rounds, host_short);
    else hipLaunchKernelGGL(select_round_kernel<false>, dim3((unsigned)blocks), dim3(SEL_THREADS), 0, st, view, s, round, rounds, host_short);
  }
}

// ------------------------------------------------------------------------------------------------
// 5. compaction in ordinal order
// ------------------------------------------------------------------------------------------------
constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS = 4;  // one 16-byte load per thread: every wave reads 1 KiB contiguous
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;

size_t compact_blocks(uint64_t M) { return (size_t)((M + CP_TILE - 1) / CP_TILE); }

template <bool SEG>
__device__ __forceinline__ void load_tile_keys(const KeyView& view, uint64_t base, uint32_t keys[CP_ITEMS], int& valid) {
  const uint32_t* __restrict__ wkey = view.base;
  const uint64_t M = view_count(view);
  if (SEG) {  // CP_ITEMS == 4 and seg_len % 4 == 0: one group of the view
    const uint4 k4 = base < M ? view_load4<true>(view, base >> 2) : make_uint4(0u, 0u, 0u, 0u);
    keys[0] = k4.x; keys[1] = k4.y; keys[2] = k4.z; keys[3] = k4.w;
    valid = base < M ? CP_ITEMS : 0;
    return;
  }
  if (base + CP_ITEMS <= M) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(wkey + base);
    keys[0] = k4.x; keys[1] = k4.y; keys[2] = k4.z; keys[3] = k4.w;
    valid = CP_ITEMS;
  } else {
    valid = base < M ? (int)(M - base) : 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++) keys[k] = (k < valid) ? wkey[base + k] : 0u;
  }
}

static_assert(CP_ITEMS == 4, "load_tile_keys reads one uint4 group of a segmented view");
template <bool SEG>
__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(KeyView view, SelectState* __restrict__ sel, int rounds,
                                                                   uint32_t* __restrict__ blk_gt,
                                                                   uint32_t* __restrict__ blk_eq,
                                                                   uint64_t* __restrict__ host_short) {
  static_assert(CP_THREADS == SEL_THREADS, "select_resolve: one workgroup of SEL_THREADS");
  __shared__ uint32_t lds[2][4];
  __shared__ uint64_t rl[10];
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE + (uint64_t)threadIdx.x * CP_ITEMS;
  uint32_t keys[CP_ITEMS];
  int valid;
  load_tile_keys<SEG>(view, base, keys, valid);
  // the last select round's histogram -> k* (every workgroup by itself; workgroup 0 leaves the result for the launches after this one)
  bool shortfall;
  const SelSnap fin = select_resolve(sel, rounds - 1, rounds, rl, &shortfall);
  const uint32_t kstar = fin.done ? fin.lo : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sel->kstar = kstar; sel->done = fin.done; sel->want = fin.want; sel->need_eq = fin.need_eq;
    if (shortfall && host_short) publish_host(host_short, 1ull);
  }
  uint32_t g = 0, q = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++)
    if (k < valid) { g += keys[k] > kstar; q += keys[k] == kstar; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { g += __shfl_xor(g, o); q += __shfl_xor(q, o); }
  if ((threadIdx.x & 63) == 0) { lds[0][threadIdx.x >> 6] = g; lds[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_gt[blockIdx.x] = lds[0][0] + lds[0][1] + lds[0][2] + lds[0][3];
    blk_eq[blockIdx.x] = lds[1][0] + lds[1][1] + lds[1][2] + lds[1][3];
  }
}

void launch_compact_count(const KeyView& view, SelectState* s, int rounds, uint32_t* blk_gt, uint32_t* blk_eq,
                          hipStream_t st, uint64_t* host_short) {
  if (view.M == 0) return;
  if (rounds > 3) rounds = 3;
  const dim3 grid((unsigned)compact_blocks(view.M));
  if (view.seg_len) hipLaunchKernelGGL(compact_count_kernel<true>, grid, dim3(CP_THREADS), 0, st, view, s, rounds, blk_gt, blk_eq, host_short);
  else hipLaunchKernelGGL(compact_count_kernel<false>, grid, dim3(CP_THREADS), 0, st, view, s, rounds, blk_gt, blk_eq, host_short);
}

template <bool SEG>
__global__ __launch_bounds__(CP_THREADS) void compact_write_kernel(KeyView view,
                                                                   SelectState* __restrict__ sel,
                                                                   const uint32_t* __restrict__ blk_gt,
                                                                   const uint32_t* __restrict__ blk_eq,
                                                                   const uint64_t* __restrict__ off_gt,
                                                                   const uint64_t* __restrict__ off_eq,
                                                                   uint64_t* __restrict__ sel_ord,
                                                                   uint32_t* __restrict__ sel_key, uint64_t n_sel) {
  // n_sel: entries sel_ord / sel_key hold.  A position at or beyond it cannot come out of a consistent select; a call that
  // was launched before the host knew the counts (and is about to be repeated) may hold anything, and must stay in bounds.
  __shared__ uint64_t lds[8];
  const uint32_t kstar = sel->kstar;
  const uint64_t need_eq = sel->need_eq;
  // the select's histograms are all zero again behind it (no launch reads them any more; a call may select twice: sharded stage B)
  for (uint32_t z = blockIdx.x * CP_THREADS + threadIdx.x; z < 3u * 4096u; z += gridDim.x * CP_THREADS) (&sel->hist[0][0])[z] = 0u;
  // off_gt == nullptr: this block adds up the tile counts before it by itself (a few thousand u32 from L2) — saves
  // the scan launch in between; both sums ride one u64 (gt high, eq low: each below 2^32 since M < 2^32 here)
  uint64_t gt0, eq0;
  if (off_gt == nullptr) {
    uint64_t a = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += CP_THREADS) a += ((uint64_t)blk_gt[b] << 32) | blk_eq[b];
    a = block_reduce_u64(a, lds);
    gt0 = a >> 32; eq0 = a & 0xFFFFFFFFull;
  } else {
    gt0 = off_gt[blockIdx.x]; eq0 = off_eq[blockIdx.x];
  }
  // most tiles hold nothing to emit (T << M): skip them on the block counts alone, without touching the keys
  if (blk_gt[blockIdx.x] == 0 && (blk_eq[blockIdx.x] == 0 || eq0 >= need_eq)) return;
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE + (uint64_t)threadIdx.x * CP_ITEMS;
  uint32_t keys[CP_ITEMS];
  int valid;
  load_tile_keys<SEG>(view, base, keys, valid);
  uint32_t g = 0, q = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++)
    if (k < valid) { g += keys[k] > kstar; q += keys[k] == kstar; }
  uint64_t tot;
  // both counts ride one u64 scan: gt in the high half, eq in the low half (each <= 1024 per tile)
  const uint64_t ex = block_exscan_u64(((uint64_t)g << 32) | q, lds, &tot);
  uint64_t gt_before = gt0 + (ex >> 32);
  uint64_t eq_before = eq0 + (ex & 0xFFFFFFFFull);
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++) {
    if (k < valid) {
      const uint32_t key = keys[k];
      const bool isg = key > kstar, isq = key == kstar;
      if (isg || (isq && eq_before < need_eq)) {
        const uint64_t pos = gt_before + (eq_before < need_eq ? eq_before : need_eq);
        if (pos < n_sel) { sel_ord[pos] = base + k; sel_key[pos] = key; }
      }
      gt_before += isg;
      eq_before += isq;
    }
  }
}

void launch_compact_write(const KeyView& view, SelectState* s, const uint32_t* blk_gt,
                          const uint32_t* blk_eq, const uint64_t* off_gt, const uint64_t* off_eq,
                          uint64_t* sel_ord, uint32_t* sel_key, uint64_t n_sel, hipStream_t st) {
  if (view.M == 0) return;
  const dim3 grid((unsigned)compact_blocks(view.M));
  if (view.seg_len)
    hipLaunchKernelGGL(compact_write_kernel<true>, grid, dim3(CP_THREADS), 0, st, view, s, blk_gt, blk_eq, off_gt, off_eq,
                       sel_ord, sel_key, n_sel);
  else
    hipLaunchKernelGGL(compact_write_kernel<false>, grid, dim3(CP_THREADS), 0, st, view, s, blk_gt, blk_eq, off_gt, off_eq,
                       sel_ord, sel_key, n_sel);
}

// ------------------------------------------------------------------------------------------------
// 5b. sharded stage B: candidate exchange
//
// A rank enumerates the triangles of ITS contiguous row range, selects its own top-T among them (the same select and
// compaction as above, on its own keys) and writes them — in (i,j,k) order — into a fixed-size record, the *blob*, that
// the caller all-gathers:  header (CAND_HDR_WORDS x u64) | keys[cap] u32 | recs[cap] uint4 {i, j, k, key}.
// The global top-T is a subset of the union of the ranks' top-T lists, and because the row ranges are contiguous and
// ascending with the rank, the concatenation of the blobs in rank order IS in global (i,j,k) order: the merge is the
// same exact select + ordinal-order compaction over the concatenated keys (a segmented KeyView), and every rank gets
// the identical selected list.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_emit_kernel(const uint64_t* __restrict__ sel_ord,
                                                        const uint32_t* __restrict__ sel_key,
                                                        const uint2* __restrict__ kcol,
                                                        const uint32_t* __restrict__ ei,
                                                        const uint32_t* __restrict__ ej,
                                                        const SelectState* __restrict__ sel,
                                                        const uint64_t* __restrict__ toff, uint64_t E,
                                                        uint64_t* __restrict__ hdr, uint32_t* __restrict__ keys,
                                                        uint4* __restrict__ recs, uint64_t cap,
                                                        uint64_t want_requested) {
  const uint64_t n_sent = sel->want < cap ? sel->want : cap;  // the select clamps want to what its window held
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[0] = toff[E];      // triangles this rank enumerated
    hdr[1] = n_sent;
    hdr[2] = sel->kstar;
    hdr[3] = sel->kmin;    // a range containing every key sent
    hdr[4] = sel->kmax;
    // cut: the rank has more triangles than it was allowed to select, and the select window held that many — so keys at
    // or below its threshold stayed behind (had the window held fewer, everything that can matter was sent)
    hdr[5] = (toff[E] > want_requested && sel->want >= want_requested) ? 1ull : 0ull;
  }
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_sent) return;
  const uint2 ke = kcol[sel_ord[p]];
  const uint32_t key = sel_key[p];
  keys[p] = key;
  recs[p] = make_uint4(ei[ke.y], ej[ke.y], ke.x, key);
}

size_t cand_cap(uint32_t T, uint32_t world, int level) {
  const size_t full = ((size_t)T + 1023) / 1024 * 1024;
  if (world <= 1) return full;
  size_t base = 2 * (((size_t)T + world - 1) / world);
  if (base < 4096) base = 4096;
  if (level > 0) base = level >= 30 ? full : base << level;
  if (level < 0) base = base >> (-level > 30 ? 30 : -level);
  size_t c = (base + 1023) / 1024 * 1024;
  if (c < 1024) c = 1024;
  return c < full ? c : full;
}
size_t cand_blob_bytes(size_t cap) { return CAND_HDR_WORDS * 8 + cap * (4 + 16); }

CandBlob cand_blob(void* blob, size_t cap) {
  CandBlob b;
  unsigned char* p = static_cast<unsigned char*>(blob);
  b.hdr = reinterpret_cast<uint64_t*>(p);
  b.cap = cap;
  b.keys = reinterpret_cast<uint32_t*>(p + CAND_HDR_WORDS * 8);
  b.recs = reinterpret_cast<uint4*>(p + CAND_HDR_WORDS * 8 + b.cap * 4);
  return b;
}

void launch_cand_emit(const uint64_t* sel_ord, const uint32_t* sel_key, const uint2* kcol, const uint32_t* ei,
                      const uint32_t* ej, const SelectState* sel, const uint64_t* toff, uint64_t E, uint32_t n_max,
                      uint64_t want_requested, const CandBlob& b, hipStream_t st) {
  const unsigned blocks = n_max ? (n_max + 255) / 256 : 1;
  hipLaunchKernelGGL(cand_emit_kernel, dim3(blocks), dim3(256), 0, st, sel_ord, sel_key, kcol, ei, ej, sel, toff, E, b.hdr,
                     b.keys, b.recs, (uint64_t)b.cap, want_requested);
}

__global__ __launch_bounds__(64) void merge_check_kernel(const uint64_t* __restrict__ blobs, uint64_t blob_words,
                                                         uint32_t world, const SelectState* __restrict__ sel,
                                                         uint64_t* __restrict__ host_flag) {
  bool bad = false;
  const uint32_t kstar = sel->kstar;
  for (uint32_t q = threadIdx.x; q < world; q += 64) {
    const uint64_t* h = blobs + (uint64_t)q * blob_words;
    if (h[5] != 0 && !(kstar > (uint32_t)h[2])) bad = true;  // a cut list whose threshold the merged one does not clear
  }
  const uint64_t any = __ballot(bad);
  if (threadIdx.x == 0) publish_host(host_flag, any ? 1ull : 0ull);
}

void launch_merge_check(const void* blobs, size_t blob_bytes, uint32_t world, const SelectState* sel, uint64_t* host_flag,
                        hipStream_t st) {
  hipLaunchKernelGGL(merge_check_kernel, dim3(1), dim3(64), 0, st, static_cast<const uint64_t*>(blobs),
                     (uint64_t)(blob_bytes / 8), world, sel, host_flag);
}

// One wave: adds up the headers of the `world` gathered blobs and arms the select state for the merge.
//   want = min(T, candidates sent);   window: [certified bound or 2.0, 3.0] when known a priori (fast != 0), else the
//   union of the ranks' key ranges.   host_out[0..1] (pinned): T_eff, triangles enumerated by all ranks.
__global__ __launch_bounds__(64) void merge_prepare_kernel(const uint64_t* __restrict__ blobs, uint64_t blob_words,
                                                           uint32_t world, uint32_t T, int fast,
                                                           const uint32_t* __restrict__ klb,
                                                           SelectState* __restrict__ sel,
                                                           uint64_t* __restrict__ host_out,
                                                           uint64_t* __restrict__ host_short) {
  const uint32_t r = threadIdx.x;
  uint64_t m = 0, ns = 0;
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
  for (uint32_t q = r; q < world; q += 64) {
    const uint64_t* h = blobs + (uint64_t)q * blob_words;
    m += h[0]; ns += h[1];
    if (h[1] != 0) { kmin = min(kmin, (uint32_t)h[3]); kmax = max(kmax, (uint32_t)h[4]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m += __shfl_xor(m, o); ns += __shfl_xor(ns, o);
    kmin = min(kmin, (uint32_t)__shfl_xor(kmin, o)); kmax = max(kmax, (uint32_t)__shfl_xor(kmax, o));
  }
  if (r != 0) return;
  const uint64_t want = ns < (uint64_t)T ? ns : (uint64_t)T;
  sel->want_req = 0;
  if (fast) {
    const uint32_t lo = *klb ? *klb : 0x40000000u, hi = 0x40400000u;  // 2.0f, 3.0f
    const uint32_t range_m1 = hi - lo;
    sel->kmin = lo; sel->kmax = hi;
    sel->st[0] = SelSnap{lo, range_m1 == 0 ? 0u : (uint32_t)(32 - __builtin_clz(range_m1)), 1u, 0u, want, 0ull, 0ull, 0ull};
  } else {
    sel->kmin = kmin <= kmax ? kmin : 0u; sel->kmax = kmin <= kmax ? kmax : 0u;
    sel->st[0] = SelSnap{0u, 0u, 0u, 0u, want, 0ull, 0ull, 0ull};
  }
  sel->want = want;  // (what the candidate kernels read if no select runs: an empty view)
  // SC_FLAG_EST_BOUND: the bound in *klb was an estimate.  A rank only ever sends keys at or above it (its select window's
  // floor), so fewer than T entries in all means fewer than T triangles of the whole graph lie above it — unless it certified
  // nothing (0: no pruning) — and the top-T of the pruned graph proves nothing about the full one (sc_tri.hip 3c)
  if (host_short && *klb != 0u && ns < (uint64_t)T) publish_host(host_short, 1ull);
  host_out[1] = m;
  publish_host(host_out, want);
}

void launch_merge_prepare(const void* blobs, size_t blob_bytes, uint32_t world, uint32_t T, bool fast, const uint32_t* klb,
                          SelectState* sel, uint64_t* host_out, hipStream_t st, uint64_t* host_short) {
  hipLaunchKernelGGL(merge_prepare_kernel, dim3(1), dim3(64), 0, st, static_cast<const uint64_t*>(blobs),
                     (uint64_t)(blob_bytes / 8), world, T, fast ? 1 : 0, klb, sel, host_out, host_short);
}

KeyView cand_view(const void* blobs, size_t blob_bytes, uint32_t world, size_t cap) {
  const unsigned char* p = static_cast<const unsigned char*>(blobs);
  KeyView v;
  v.base = reinterpret_cast<const uint32_t*>(p + CAND_HDR_WORDS * 8);
  v.seg_len = cap;
  v.M = (uint64_t)world * v.seg_len;
  v.seg_stride = blob_bytes / 4;
  v.valid = reinterpret_cast<const uint64_t*>(p) + 1;  // hdr[1] = entries sent
  v.valid_stride = blob_bytes / 8;
  v.M_dev = nullptr;
  return v;
}

// ------------------------------------------------------------------------------------------------
// 7. decode: selected position -> ordinal -> edge (binary search in toff) -> r-th common neighbour above j.
//    The list stays in ORDINAL order ((i,j,k) ascending): neighbouring threads hit neighbouring edges, and the hot
//    path never needs the ranked order (see score_argmax_kernel).  launch_rank_order produces it for the stage hook.
// ------------------------------------------------------------------------------------------------
// kcol[ordinal] = {third vertex, edge id}, stored by the key kernels: decode is two lookups, no search (the earlier
// form searched toff for the edge: 12 LDS + 7 global steps per position, 10 us at T = 50 k and 28 us at 400 k).
__global__ __launch_bounds__(256) void tri_decode_kernel(const uint32_t* __restrict__ ei,
                                                         const uint32_t* __restrict__ ej,
                                                         const uint2* __restrict__ kcol,
                                                         const uint64_t* __restrict__ sel_ord, uint32_t T,
                                                         uint32_t* __restrict__ tri) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const uint2 ke = kcol[sel_ord[t]];
  tri[3 * (size_t)t] = ei[ke.y];
  tri[3 * (size_t)t + 1] = ej[ke.y];
  tri[3 * (size_t)t + 2] = ke.x;
}

void launch_tri_decode(const uint32_t* ei, const uint32_t* ej, const uint2* kcol, const uint64_t* sel_ord, uint32_t T,
                       uint32_t* tri, hipStream_t st) {
  if (T == 0) return;
  hipLaunchKernelGGL(tri_decode_kernel, dim3((T + 255) / 256), dim3(256), 0, st, ei, ej, kcol, sel_ord, T, tri);
}

// ---- ranked order for the stage hook: sort (~key, position), then gather
__global__ __launch_bounds__(256) void sortkey_kernel(const uint32_t* __restrict__ sel_key, uint32_t T,
                                                      uint64_t* __restrict__ sortkey) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) sortkey[t] = ((uint64_t)(~sel_key[t]) << 32) | (uint64_t)t;
}
__global__ __launch_bounds__(256) void gather_ranked_kernel(const uint64_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ tri,
                                                            const uint32_t* __restrict__ sel_key, uint32_t T,
                                                            uint32_t* __restrict__ tri_ranked,
                                                            uint32_t* __restrict__ key_ranked) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const uint32_t src = (uint32_t)(sorted[t] & 0xFFFFFFFFull);
  tri_ranked[3 * (size_t)t] = tri[3 * (size_t)src];
  tri_ranked[3 * (size_t)t + 1] = tri[3 * (size_t)src + 1];
  tri_ranked[3 * (size_t)t + 2] = tri[3 * (size_t)src + 2];
  key_ranked[t] = sel_key[src];
}

void launch_rank_order(const uint32_t* tri, const uint32_t* sel_key, uint32_t T, uint64_t* sortkey, uint64_t* sorted,
                       void* sort_tmp, size_t sort_bytes, uint32_t* tri_ranked, uint32_t* key_ranked,
                       hipStream_t st) {
  if (T == 0) return;
  hipLaunchKernelGGL(sortkey_kernel, dim3((T + 255) / 256), dim3(256), 0, st, sel_key, T, sortkey);
  launch_sort_u64(sortkey, sorted, T, sort_tmp, sort_bytes, st);
  hipLaunchKernelGGL(gather_ranked_kernel, dim3((T + 255) / 256), dim3(256), 0, st, sorted, tri, sel_key, T,
                     tri_ranked, key_ranked);
}

}  // namespace sc
