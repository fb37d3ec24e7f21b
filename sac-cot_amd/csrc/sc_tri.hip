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
#include "sc_arith.hpp"
#include "sc_block.hpp"
#include "sc_kernels.hpp"

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

// ------------------------------------------------------------------------------------------------
// 1. edge_fill: one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_fill_kernel(const uint64_t* __restrict__ bits,
                                                        const float* __restrict__ S, int n, int ld, int W,
                                                        const uint64_t* __restrict__ edge_off,
                                                        uint32_t* __restrict__ ei, uint32_t* __restrict__ ej,
                                                        float* __restrict__ es) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  uint64_t base = edge_off[i];
  const int w0 = i >> 6;
  for (int wb = w0; wb < W; wb += 64) {
    const int w = wb + lane;
    uint64_t v = 0;
    if (w < W) {
      v = bits[(size_t)i * W + w];
      if (w == w0) v &= mask_above(i & 63);
    }
    uint32_t tot;
    uint32_t r = group_exscan<64>((uint32_t)__popcll(v), &tot);
    while (v) {
      const int b = __builtin_ctzll(v);
      v &= v - 1;
      const uint32_t j = (uint32_t)(w * 64 + b);
      const uint64_t e = base + r++;
      ei[e] = (uint32_t)i;
      ej[e] = j;
      es[e] = S[(size_t)i * ld + j];
    }
    base += tot;
  }
}

void launch_edge_fill(const Graph& g, const uint64_t* edge_off, uint32_t* ei, uint32_t* ej, float* es,
                      hipStream_t st) {
  hipLaunchKernelGGL(edge_fill_kernel, dim3((g.n + 3) / 4), dim3(256), 0, st, g.bits, g.S, g.n, g.ld, g.W,
                     edge_off, ei, ej, es);
}

// ------------------------------------------------------------------------------------------------
// 2. tri_count: 16 lanes per edge
// ------------------------------------------------------------------------------------------------
constexpr int TG = 16;  // lanes per edge

__global__ __launch_bounds__(256) void tri_count_kernel(const uint64_t* __restrict__ bits, int W,
                                                        const uint32_t* __restrict__ ei,
                                                        const uint32_t* __restrict__ ej, uint64_t E,
                                                        uint32_t* __restrict__ tcnt) {
  const int gl = threadIdx.x & (TG - 1);
  const uint64_t e = (uint64_t)blockIdx.x * (256 / TG) + (threadIdx.x / TG);
  const bool live = e < E;
  const uint64_t ec = live ? e : 0;
  const uint32_t i = ei[ec], j = ej[ec];
  const uint64_t* ri = bits + (size_t)i * W;
  const uint64_t* rj = bits + (size_t)j * W;
  const int w0 = j >> 6;
  uint32_t c = 0;
  if (live) {
    for (int w = w0 + gl; w < W; w += TG) {
      uint64_t m = ri[w] & rj[w];
      if (w == w0) m &= mask_above(j & 63);
      c += __popcll(m);
    }
  }
#pragma unroll
  for (int o = TG / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, TG);
  if (live && gl == 0) tcnt[e] = c;
}

void launch_tri_count(const Graph& g, const uint32_t* ei, const uint32_t* ej, uint64_t E, uint32_t* tcnt,
                      hipStream_t st) {
  if (E == 0) return;
  const uint64_t per = 256 / TG;
  hipLaunchKernelGGL(tri_count_kernel, dim3((unsigned)((E + per - 1) / per)), dim3(256), 0, st, g.bits, g.W, ei, ej,
                     E, tcnt);
}

// ------------------------------------------------------------------------------------------------
// 3. tri_keys
// ------------------------------------------------------------------------------------------------
__global__ void select_init_kernel(SelectState* s, uint64_t want) {
  for (int b = threadIdx.x; b < 2048; b += blockDim.x) s->hist[b] = 0;
  if (threadIdx.x == 0) {
    s->kmin = 0xFFFFFFFFu; s->kmax = 0; s->lo = 0; s->wbits = 0xFFFFFFFFu; s->done = 0; s->kstar = 0;
    s->want = want; s->above = 0; s->need_eq = 0;
  }
}
void launch_select_init(SelectState* s, uint64_t want, hipStream_t st) {
  hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, st, s, want);
}

constexpr int TK_MAX_BLOCKS = 8192;  // bounds the per-block min/max arrays

__global__ __launch_bounds__(256) void tri_keys_kernel(const uint64_t* __restrict__ bits, int W,
                                                       const uint32_t* __restrict__ deg,
                                                       const uint64_t* __restrict__ edge_off,
                                                       const uint32_t* __restrict__ ei,
                                                       const uint32_t* __restrict__ ej,
                                                       const float* __restrict__ es,
                                                       const uint64_t* __restrict__ toff, uint64_t E,
                                                       int rank_mode, uint32_t* __restrict__ wkey,
                                                       uint32_t* __restrict__ blk_min,
                                                       uint32_t* __restrict__ blk_max) {
  __shared__ uint32_t lmin[4], lmax[4];
  const int gl = threadIdx.x & (TG - 1);
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
  const uint64_t groups = (uint64_t)gridDim.x * (256 / TG);
  // every group of TG lanes walks edges e, e + groups, ...; all lanes of a group share the trip counts
  for (uint64_t e = (uint64_t)blockIdx.x * (256 / TG) + (threadIdx.x / TG); e < E; e += groups) {
    const uint32_t i = ei[e], j = ej[e];
    const uint64_t* ri = bits + (size_t)i * W;
    const uint64_t* rj = bits + (size_t)j * W;
    const int w0 = j >> 6;
    const float s_ij = es[e];
    const uint32_t dsum_ij = deg[i] + deg[j];
    const uint64_t out0 = toff[e];
    const uint64_t eik0 = e + 1;           // edge index of (i, first neighbour of i above j)
    const uint64_t ejk0 = edge_off[j];     // edge index of (j, first neighbour of j above j)
    uint32_t base_m = 0, base_i = 0, base_j = 0;
    const int rounds = (W - w0 + TG - 1) / TG;
    for (int it = 0; it < rounds; it++) {
      const int w = w0 + it * TG + gl;
      uint64_t ai = 0, aj = 0;
      if (w < W) {
        ai = ri[w]; aj = rj[w];
        if (w == w0) { const uint64_t mk = mask_above(j & 63); ai &= mk; aj &= mk; }
      }
      uint64_t m = ai & aj;
      uint32_t tm, ti, tj;
      uint32_t pm = base_m + group_exscan<TG>((uint32_t)__popcll(m), &tm);
      const uint32_t pi = base_i + group_exscan<TG>((uint32_t)__popcll(ai), &ti);
      const uint32_t pj = base_j + group_exscan<TG>((uint32_t)__popcll(aj), &tj);
      base_m += tm; base_i += ti; base_j += tj;
      while (m) {
        const int b = __builtin_ctzll(m);
        const uint64_t below = (1ull << b) - 1ull;
        m &= m - 1;
        uint32_t key;
        if (rank_mode == 0) {
          const float s_ik = es[eik0 + pi + (uint32_t)__popcll(ai & below)];
          const float s_jk = es[ejk0 + pj + (uint32_t)__popcll(aj & below)];
          key = __float_as_uint((s_ij + s_ik) + s_jk);
        } else {
          key = dsum_ij + deg[w * 64 + b];
        }
        wkey[out0 + pm++] = key;
        kmin = min(kmin, key);
        kmax = max(kmax, key);
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
                                                         SelectState* __restrict__ sel) {
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
  }
}

size_t tri_keys_blocks(uint64_t E) {
  const uint64_t per = 256 / TG;
  const uint64_t nb = (E + per - 1) / per;
  return (size_t)(nb < (uint64_t)TK_MAX_BLOCKS ? nb : (uint64_t)TK_MAX_BLOCKS);
}

void launch_tri_keys(const Graph& g, const uint64_t* edge_off, const uint32_t* ei, const uint32_t* ej,
                     const float* es, const uint64_t* toff, uint64_t E, int rank_mode, uint32_t* wkey,
                     uint32_t* blk_minmax, SelectState* s, hipStream_t st) {
  if (E == 0) return;
  const int nb = (int)tri_keys_blocks(E);
  hipLaunchKernelGGL(tri_keys_kernel, dim3(nb), dim3(256), 0, st, g.bits, g.W, g.deg, edge_off, ei, ej, es, toff, E,
                     rank_mode, wkey, blk_minmax, blk_minmax + TK_MAX_BLOCKS);
  hipLaunchKernelGGL(key_range_kernel, dim3(1), dim3(1024), 0, st, blk_minmax, blk_minmax + TK_MAX_BLOCKS, nb, s);
}

// ------------------------------------------------------------------------------------------------
// 4. radix select: window [lo, lo + 2048 << shift), <= 3 rounds down to shift 0
// ------------------------------------------------------------------------------------------------
constexpr int SEL_BINS = 2048;
constexpr int SEL_THREADS = 256;
constexpr int SEL_ITEMS = 16;

// current window of the select: keys in [lo, lo + 2^wbits), binned by (key - lo) >> shift into <= 2048 bins
struct SelWindow { uint32_t lo, wbits, shift; };
__device__ __forceinline__ SelWindow select_window(const SelectState* sel) {
  SelWindow w;
  if (sel->wbits == 0xFFFFFFFFu) {  // first round: the window is the key range [kmin, kmax]
    w.lo = sel->kmin;
    const uint32_t range_m1 = sel->kmax - sel->kmin;
    w.wbits = range_m1 == 0 ? 0u : (uint32_t)(32 - __builtin_clz(range_m1));
  } else {
    w.lo = sel->lo;
    w.wbits = sel->wbits;
  }
  w.shift = w.wbits > 11 ? w.wbits - 11 : 0u;
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

__global__ __launch_bounds__(SEL_THREADS) void select_hist_kernel(const uint32_t* __restrict__ wkey, uint64_t M,
                                                                  SelectState* __restrict__ sel) {
  __shared__ uint32_t lh[SEL_BINS];
  if (sel->done) return;
  for (int b = threadIdx.x; b < SEL_BINS; b += SEL_THREADS) lh[b] = 0;
  __syncthreads();
  const SelWindow win = select_window(sel);
  const uint64_t width = 1ull << win.wbits;
  const uint64_t M4 = M >> 2;  // whole uint4 groups (wkey comes from hipMalloc: 16-byte aligned)
  const uint4* __restrict__ wkey4 = reinterpret_cast<const uint4*>(wkey);
  const uint64_t stride = (uint64_t)gridDim.x * SEL_THREADS;
  for (uint64_t q = (uint64_t)blockIdx.x * SEL_THREADS + threadIdx.x; q < M4; q += stride) {
    const uint4 k4 = wkey4[q];
    const uint32_t ks[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint64_t rel = (uint64_t)ks[c] - (uint64_t)win.lo;  // wraps huge when key < lo
      hist_add(lh, (ks[c] >= win.lo) && (rel < width), (uint32_t)(rel >> win.shift));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (M & 3)) {  // tail
    const uint32_t key = wkey[(M4 << 2) + threadIdx.x];
    const uint64_t rel = (uint64_t)key - (uint64_t)win.lo;
    if ((key >= win.lo) && (rel < width)) atomicAdd(&lh[(uint32_t)(rel >> win.shift)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SEL_BINS; b += SEL_THREADS) {
    const uint32_t v = lh[b];
    if (v) atomicAdd(&sel->hist[b], v);
  }
}

// one block: walk the bins from the top, pick the bin holding the want-th key, narrow the window
__global__ __launch_bounds__(256) void select_pick_kernel(SelectState* __restrict__ sel) {
  __shared__ uint64_t lds[8];
  __shared__ uint32_t s_bin;
  __shared__ uint64_t s_above;
  if (sel->done) return;
  const SelWindow win = select_window(sel);
  const uint64_t want = sel->want, above0 = sel->above;
  // thread t owns bins [8t, 8t+8) counted from the TOP: bin index = 2047 - (8t + k)
  uint32_t h[8];
  uint64_t mine = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { h[k] = sel->hist[SEL_BINS - 1 - (threadIdx.x * 8 + k)]; mine += h[k]; }
  if (threadIdx.x == 0) { s_bin = 0; s_above = above0; }
  uint64_t tot;
  const uint64_t before = above0 + block_exscan_u64(mine, lds, &tot);
  // the crossing thread: before < want <= before + mine (exactly one: the window holds >= want - above0 keys)
  if (before < want && want <= before + mine) {
    uint64_t run = before;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (run < want && want <= run + h[k]) { s_bin = SEL_BINS - 1 - (threadIdx.x * 8 + k); s_above = run; }
      run += h[k];
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) sel->hist[b] = 0;
  if (threadIdx.x == 0) {
    const uint32_t nlo = win.lo + (s_bin << win.shift);
    sel->above = s_above;
    sel->lo = nlo;
    sel->wbits = win.shift;  // the chosen bin is the next window
    if (win.shift == 0) { sel->done = 1; sel->kstar = nlo; sel->need_eq = want - s_above; }
  }
}

void launch_select_rounds(const uint32_t* wkey, uint64_t M, SelectState* s, hipStream_t st) {
  if (M == 0) return;
  uint64_t blocks = (M + (uint64_t)SEL_THREADS * SEL_ITEMS - 1) / ((uint64_t)SEL_THREADS * SEL_ITEMS);
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  for (int round = 0; round < 3; round++) {
    hipLaunchKernelGGL(select_hist_kernel, dim3((unsigned)blocks), dim3(SEL_THREADS), 0, st, wkey, M, s);
    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, st, s);
  }
}

// ------------------------------------------------------------------------------------------------
// 5. compaction in ordinal order
// ------------------------------------------------------------------------------------------------
constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS = 4;  // one 16-byte load per thread: every wave reads 1 KiB contiguous
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;

size_t compact_blocks(uint64_t M) { return (size_t)((M + CP_TILE - 1) / CP_TILE); }

__device__ __forceinline__ void load_tile_keys(const uint32_t* __restrict__ wkey, uint64_t M, uint64_t base,
                                               uint32_t keys[CP_ITEMS], int& valid) {
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

__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(const uint32_t* __restrict__ wkey, uint64_t M,
                                                                   const SelectState* __restrict__ sel,
                                                                   uint32_t* __restrict__ blk_gt,
                                                                   uint32_t* __restrict__ blk_eq) {
  __shared__ uint32_t lds[2][4];
  const uint32_t kstar = sel->kstar;
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE + (uint64_t)threadIdx.x * CP_ITEMS;
  uint32_t keys[CP_ITEMS];
  int valid;
  load_tile_keys(wkey, M, base, keys, valid);
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

void launch_compact_count(const uint32_t* wkey, uint64_t M, const SelectState* s, uint32_t* blk_gt,
                          uint32_t* blk_eq, hipStream_t st) {
  if (M == 0) return;
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)compact_blocks(M)), dim3(CP_THREADS), 0, st, wkey, M, s,
                     blk_gt, blk_eq);
}

__global__ __launch_bounds__(CP_THREADS) void compact_write_kernel(const uint32_t* __restrict__ wkey, uint64_t M,
                                                                   const SelectState* __restrict__ sel,
                                                                   const uint32_t* __restrict__ blk_gt,
                                                                   const uint32_t* __restrict__ blk_eq,
                                                                   const uint64_t* __restrict__ off_gt,
                                                                   const uint64_t* __restrict__ off_eq,
                                                                   uint64_t* __restrict__ sel_ord,
                                                                   uint64_t* __restrict__ sortkey) {
  __shared__ uint64_t lds[8];
  const uint32_t kstar = sel->kstar;
  const uint64_t need_eq = sel->need_eq;
  // most tiles hold nothing to emit (T << M): skip them on the block counts alone, without touching the keys
  const uint64_t eq0 = off_eq[blockIdx.x];
  if (blk_gt[blockIdx.x] == 0 && (blk_eq[blockIdx.x] == 0 || eq0 >= need_eq)) return;
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE + (uint64_t)threadIdx.x * CP_ITEMS;
  uint32_t keys[CP_ITEMS];
  int valid;
  load_tile_keys(wkey, M, base, keys, valid);
  uint32_t g = 0, q = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++)
    if (k < valid) { g += keys[k] > kstar; q += keys[k] == kstar; }
  uint64_t tot;
  // both counts ride one u64 scan: gt in the high half, eq in the low half (each <= 1024 per tile)
  const uint64_t ex = block_exscan_u64(((uint64_t)g << 32) | q, lds, &tot);
  uint64_t gt_before = off_gt[blockIdx.x] + (ex >> 32);
  uint64_t eq_before = eq0 + (ex & 0xFFFFFFFFull);
#pragma unroll
  for (int k = 0; k < CP_ITEMS; k++) {
    if (k < valid) {
      const uint32_t key = keys[k];
      const bool isg = key > kstar, isq = key == kstar;
      if (isg || (isq && eq_before < need_eq)) {
        const uint64_t pos = gt_before + (eq_before < need_eq ? eq_before : need_eq);
        sel_ord[pos] = base + k;
        sortkey[pos] = ((uint64_t)(~key) << 32) | (uint64_t)(uint32_t)pos;
      }
      gt_before += isg;
      eq_before += isq;
    }
  }
}

void launch_compact_write(const uint32_t* wkey, uint64_t M, const SelectState* s, const uint32_t* blk_gt,
                          const uint32_t* blk_eq, const uint64_t* off_gt, const uint64_t* off_eq,
                          uint64_t* sel_ord, uint64_t* sortkey, hipStream_t st) {
  if (M == 0) return;
  hipLaunchKernelGGL(compact_write_kernel, dim3((unsigned)compact_blocks(M)), dim3(CP_THREADS), 0, st, wkey, M, s,
                     blk_gt, blk_eq, off_gt, off_eq, sel_ord, sortkey);
}

// ------------------------------------------------------------------------------------------------
// 7. decode: ranked position -> ordinal -> edge (binary search in toff) -> r-th common neighbour above j
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tri_decode_kernel(const uint64_t* __restrict__ bits, int W,
                                                         const uint32_t* __restrict__ ei,
                                                         const uint32_t* __restrict__ ej,
                                                         const uint64_t* __restrict__ toff, uint64_t E,
                                                         const uint64_t* __restrict__ sorted,
                                                         const uint64_t* __restrict__ sel_ord, uint32_t T,
                                                         uint32_t* __restrict__ tri, uint32_t* __restrict__ key) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const uint64_t sk = sorted[t];
  const uint32_t pos = (uint32_t)(sk & 0xFFFFFFFFull);
  const uint64_t ord = sel_ord[pos];
  // largest e with toff[e] <= ord  (toff has E+1 entries, toff[E] = M > ord)
  uint64_t lo = 0, hi = E;
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (toff[mid] <= ord) lo = mid; else hi = mid;
  }
  const uint64_t e = lo;
  uint32_t r = (uint32_t)(ord - toff[e]);
  const uint32_t i = ei[e], j = ej[e];
  const uint64_t* ri = bits + (size_t)i * W;
  const uint64_t* rj = bits + (size_t)j * W;
  uint32_t k = 0xFFFFFFFFu;
  for (int w = j >> 6; w < W; w++) {
    uint64_t m = ri[w] & rj[w];
    if (w == (int)(j >> 6)) m &= mask_above(j & 63);
    const uint32_t c = (uint32_t)__popcll(m);
    if (r < c) {
      for (uint32_t q = 0; q < r; q++) m &= m - 1;
      k = (uint32_t)(w * 64 + __builtin_ctzll(m));
      break;
    }
    r -= c;
  }
  tri[3 * (size_t)t] = i;
  tri[3 * (size_t)t + 1] = j;
  tri[3 * (size_t)t + 2] = k;
  key[t] = ~(uint32_t)(sk >> 32);
}

void launch_tri_decode(const Graph& g, const uint32_t* ei, const uint32_t* ej, const uint64_t* toff, uint64_t E,
                       const uint64_t* sorted, const uint64_t* sel_ord, uint32_t T, uint32_t* tri,
                       uint32_t* key, hipStream_t st) {
  if (T == 0) return;
  hipLaunchKernelGGL(tri_decode_kernel, dim3((T + 255) / 256), dim3(256), 0, st, g.bits, g.W, ei, ej, toff, E,
                     sorted, sel_ord, T, tri, key);
}

}  // namespace sc
