// sc_compat.hip — input staging, stage A (compat_graph) and the exclusive scan.
//
// Stage A is HBM-write bound: 4 N^2 bytes of weights + N^2/8 bytes of adjacency bits (103 MB at N = 5000).
// Layout: one workgroup owns COMPAT_ROWS full rows of the matrix.  Lane l of wave w owns column
// c0 + 64 w + l of each 256-column step, so
//   * the six coordinates of column j are loaded once (coalesced) and reused for COMPAT_ROWS rows,
//   * the row operands are wave-uniform (scalar loads),
//   * every store instruction of a wave writes 256 contiguous bytes of one row of S,
//   * __ballot() of the edge predicate IS the 64-bit adjacency word of that row — no shuffles,
//   * the bit rows are assembled in LDS and written once, coalesced; deg / deg+ fall out of them.
#include "sc_arith.hpp"
#include "sc_block.hpp"
#include "sc_kernels.hpp"

namespace sc {

// ------------------------------------------------------------------------------------------------
// input staging: user layout -> 6 zero-padded planes, plus a finiteness check
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage_points_kernel(const float* __restrict__ src,
                                                           const float* __restrict__ tgt, int n, int ld,
                                                           int layout, float* __restrict__ planes,
                                                           uint32_t* __restrict__ bad_flag) {
  int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= ld) return;
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (m < n) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      size_t idx = layout ? ((size_t)c * n + m) : ((size_t)m * 3 + c);
      v[c] = src[idx];
      v[3 + c] = tgt[idx];
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; c++) ok = ok && (fabsf(v[c]) < __builtin_inff());
    if (!ok) atomicOr(bad_flag, 1u);
  }
#pragma unroll
  for (int c = 0; c < 6; c++) planes[(size_t)c * ld + m] = v[c];
}

void launch_stage_points(const float* d_src, const float* d_tgt, int n, int ld, int layout, float* planes,
                         uint32_t* bad_flag, hipStream_t st) {
  hipLaunchKernelGGL(stage_points_kernel, dim3((ld + 255) / 256), dim3(256), 0, st, d_src, d_tgt, n, ld, layout,
                     planes, bad_flag);
}

// ------------------------------------------------------------------------------------------------
// stage A
// ------------------------------------------------------------------------------------------------
constexpr int COMPAT_ROWS = 4;     // rows per workgroup (= waves per workgroup, used by the epilogue)
constexpr int COMPAT_THREADS = 256;

__global__ __launch_bounds__(COMPAT_THREADS) void compat_rows_kernel(const float* __restrict__ planes, int n,
                                                                     int ld, float d_thr, float min_len,
                                                                     float nis, float* __restrict__ S,
                                                                     uint64_t* __restrict__ bits,
                                                                     uint32_t* __restrict__ deg,
                                                                     uint32_t* __restrict__ degp,
                                                                     uint32_t* __restrict__ wpre) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* lbits = reinterpret_cast<uint64_t*>(smem);  // COMPAT_ROWS x W
  const int W = ld >> 6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * COMPAT_ROWS;
  const float* px = planes;
  const float* py = planes + ld;
  const float* pz = planes + 2 * (size_t)ld;
  const float* qx = planes + 3 * (size_t)ld;
  const float* qy = planes + 4 * (size_t)ld;
  const float* qz = planes + 5 * (size_t)ld;

  // wave-uniform row operands (i0 + r < ld always: ld >= n rounded up to 64 and COMPAT_ROWS divides 64)
  float rpx[COMPAT_ROWS], rpy[COMPAT_ROWS], rpz[COMPAT_ROWS], rqx[COMPAT_ROWS], rqy[COMPAT_ROWS], rqz[COMPAT_ROWS];
#pragma unroll
  for (int r = 0; r < COMPAT_ROWS; r++) {
    const int i = i0 + r;
    rpx[r] = px[i]; rpy[r] = py[i]; rpz[r] = pz[i];
    rqx[r] = qx[i]; rqy[r] = qy[i]; rqz[r] = qz[i];
  }

  for (int c0 = 0; c0 < ld; c0 += COMPAT_THREADS) {
    const int j = c0 + threadIdx.x;
    if (c0 + wave * 64 < ld) {  // wave-uniform: ld is a multiple of 64
      const float jpx = px[j], jpy = py[j], jpz = pz[j], jqx = qx[j], jqy = qy[j], jqz = qz[j];
#pragma unroll
      for (int r = 0; r < COMPAT_ROWS; r++) {
        const int i = i0 + r;
        const float dp = dist3(rpx[r], rpy[r], rpz[r], jpx, jpy, jpz);
        const float dq = dist3(rqx[r], rqy[r], rqz[r], jqx, jqy, jqz);
        const float d = fabsf(dp - dq);
        const bool e = (d <= d_thr) && (dp >= min_len) && (dq >= min_len) && (j != i) && (j < n) && (i < n);
        const uint64_t word = __ballot(e);
        float s = 0.0f;
        if (word != 0) {  // wave-uniform: skip the polynomial when no lane holds an edge
          const float v = sc_expf((d * d) * nis);
          s = e ? v : 0.0f;
        }
        if (i < n) S[(size_t)i * ld + j] = s;
        if (lane == 0) lbits[r * W + (c0 >> 6) + wave] = word;
      }
    }
  }
  __syncthreads();
  // epilogue: wave r owns row i0 + r — write its bit row coalesced, reduce deg and deg+ (bits above i), and emit
  // the word-prefix popcounts wpre[i][w] = #set bits of row i in words [0, w): they turn "index of edge (i,k) in
  // the CSR arrays" into an O(1) lookup for stage B.
  {
    const int r = wave, i = i0 + r;
    if (i < n) {
      uint32_t d_all = 0, d_up = 0;
      for (int wb = 0; wb < W; wb += 64) {
        const int w = wb + lane;
        const uint64_t v = w < W ? lbits[r * W + w] : 0ull;
        const uint32_t pc = (uint32_t)__popcll(v);
        uint32_t inc = pc;  // inclusive wave scan of the word popcounts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_up(inc, o);
          if (lane >= o) inc += t;
        }
        if (w < W) {
          bits[(size_t)i * W + w] = v;
          wpre[(size_t)i * W + w] = d_all + inc - pc;
          uint64_t up = v;
          if (w < (i >> 6)) up = 0;
          else if (w == (i >> 6)) up &= ((i & 63) == 63) ? 0ull : (~0ull << ((i & 63) + 1));
          d_up += __popcll(up);
        }
        d_all += __shfl(inc, 63);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) d_up += __shfl_xor(d_up, o);
      if (lane == 0) { deg[i] = d_all; degp[i] = d_up; }
    }
  }
}

void launch_compat(const Points& pts, const Derived& dv, float* S, uint64_t* bits, uint32_t* deg,
                   uint32_t* degp, uint32_t* wpre, hipStream_t st) {
  static_assert(COMPAT_ROWS * 64 == COMPAT_THREADS, "epilogue maps wave r to row r");
  const int W = pts.ld >> 6;
  const int grid = (pts.n + COMPAT_ROWS - 1) / COMPAT_ROWS;
  const size_t lds = (size_t)COMPAT_ROWS * W * sizeof(uint64_t);
  hipLaunchKernelGGL(compat_rows_kernel, dim3(grid), dim3(COMPAT_THREADS), lds, st, pts.planes, pts.n, pts.ld,
                     dv.d_thr, dv.min_len, dv.neg_inv2sig2, S, bits, deg, degp, wpre);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64, three launches: block sums, scan of block sums (one block), down-sweep
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                       uint64_t* __restrict__ bsum) {
  __shared__ uint64_t lds[8];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
  s = block_reduce_u64(s, lds);
  if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void scan_of_sums_kernel(uint64_t* __restrict__ bsum, size_t nb,
                                                            uint64_t* __restrict__ total_out) {
  __shared__ uint64_t lds[16];
  uint64_t carry = 0;
  for (size_t b0 = 0; b0 < nb; b0 += 1024) {
    const size_t b = b0 + threadIdx.x;
    const uint64_t v = b < nb ? bsum[b] : 0;
    uint64_t tot;
    const uint64_t ex = block_exscan_u64(v, lds, &tot);
    if (b < nb) bsum[b] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                      const uint64_t* __restrict__ bsum,
                                                                      uint64_t* __restrict__ out) {
  __shared__ uint64_t lds[8];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? in[base + k] : 0u; s += v[k]; }
  uint64_t tot;
  uint64_t run = bsum[blockIdx.x] + block_exscan_u64(s, lds, &tot);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

// small inputs: ONE block scans up to two arrays in one launch (three launches of the tiled scan are pure latency there)
__global__ __launch_bounds__(1024) void scan_small_kernel(const uint32_t* __restrict__ in0, uint64_t* __restrict__ out0,
                                                          const uint32_t* __restrict__ in1, uint64_t* __restrict__ out1,
                                                          size_t n) {
  __shared__ uint64_t lds[16];
  const int arrays = in1 ? 2 : 1;
  for (int a = 0; a < arrays; a++) {
    const uint32_t* in = a ? in1 : in0;
    uint64_t* out = a ? out1 : out0;
    uint64_t carry = 0;
    for (size_t b0 = 0; b0 < n; b0 += 1024) {
      const size_t b = b0 + threadIdx.x;
      const uint64_t v = b < n ? in[b] : 0;
      uint64_t tot;
      const uint64_t ex = block_exscan_u64(v, lds, &tot);
      if (b < n) out[b] = carry + ex;
      carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
  }
}

constexpr size_t SCAN_SMALL_MAX = 32768;

void launch_scan_u32_pair(const uint32_t* in0, uint64_t* out0, const uint32_t* in1, uint64_t* out1, size_t n,
                          void* temp, hipStream_t st) {
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in0, out0, in1, out1, n);
  } else {
    launch_scan_u32(in0, n, out0, temp, st);
    if (in1) launch_scan_u32(in1, n, out1, temp, st);
  }
}

size_t scan_temp_bytes(size_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint64_t); }

void launch_scan_u32(const uint32_t* in, size_t n, uint64_t* out, void* temp, hipStream_t st) {
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in, out, (const uint32_t*)nullptr,
                       (uint64_t*)nullptr, n);
    return;
  }
  uint64_t* bsum = static_cast<uint64_t*>(temp);
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) { (void)hipMemsetAsync(out, 0, sizeof(uint64_t), st); return; }
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, bsum);
  hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(1024), 0, st, bsum, nb, out + n);
  hipLaunchKernelGGL(scan_downsweep_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, bsum, out);
}

}  // namespace sc
