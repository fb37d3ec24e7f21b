// sc_sort.hip — ranked order of the T selected triangles for the stage hook sc_triangles_host: ascending sort of
// (~key << 32 | position) == (key descending, ordinal ascending).  Off the hot path (the hot path never sorts).
//
// A stable LSD radix sort on the HIGH 32 bits, 8 bits per pass: the low halves are the input positions, already
// ascending, so a stable sort of the high halves IS the 64-bit sort.  Hand-written (round 1 used rocPRIM's device
// radix sort, whose headers make the library read an environment variable: the shipped .so reads none).
// Per pass: per-tile digit histograms -> exclusive scan in (digit, tile) order -> one wave per tile scatters its
// elements chunk by chunk; a lane's rank inside a chunk is the number of lower lanes with the same digit (eight
// ballots), so the order of equal digits is preserved.  Nothing depends on atomics' arrival order.
#include "sc_kernels.hpp"

namespace sc {

constexpr int RS_TILE = 2048;

__global__ __launch_bounds__(256) void rs_hist_kernel(const uint64_t* __restrict__ in, size_t n, int shift,
                                                      uint32_t* __restrict__ bh, uint32_t nb) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int k = 0; k < RS_TILE / 256; k++) {
    const size_t x = base + (size_t)k * 256 + threadIdx.x;
    if (x < n) atomicAdd(&h[(uint32_t)(in[x] >> shift) & 255u], 1u);  // integer sums: order-free
  }
  __syncthreads();
  bh[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(64) void rs_scatter_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                        size_t n, int shift, const uint64_t* __restrict__ off,
                                                        uint32_t nb) {
  __shared__ uint64_t run[256];  // next output slot of every digit for this tile
  const int lane = threadIdx.x;
  for (int b = lane; b < 256; b += 64) run[b] = off[(size_t)b * nb + blockIdx.x];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wave: LDS is in-order, the fence pins the compiler
  const size_t base = (size_t)blockIdx.x * RS_TILE;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int c = 0; c < RS_TILE / 64; c++) {
    const size_t x = base + (size_t)c * 64 + lane;
    const bool v = x < n;
    const uint64_t key = v ? in[x] : 0ull;
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    uint64_t peers = __ballot(v);
    if (peers == 0) break;  // wave-uniform: past the end
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t r = (uint32_t)__popcll(peers & lt), tot = (uint32_t)__popcll(peers);
    const uint64_t dst = run[d] + r;
    if (v) out[dst] = key;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (v && r + 1 == tot) run[d] = dst + 1;  // the last lane of every digit group advances its slot
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

static size_t rs_tiles(size_t n) { return (n + RS_TILE - 1) / RS_TILE; }
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t sort_temp_bytes(size_t n) {
  if (n == 0) return 0;
  const size_t cells = 256 * rs_tiles(n);
  return align256(n * 8) + align256(cells * 4) + align256((cells + 1) * 8) + align256(scan_temp_bytes(cells));
}

void launch_sort_u64(const uint64_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes, hipStream_t st) {
  if (n == 0 || temp_bytes < sort_temp_bytes(n)) return;
  const uint32_t nb = (uint32_t)rs_tiles(n);
  const size_t cells = 256 * (size_t)nb;
  unsigned char* p = static_cast<unsigned char*>(temp);
  uint64_t* pong = reinterpret_cast<uint64_t*>(p); p += align256(n * 8);
  uint32_t* bh = reinterpret_cast<uint32_t*>(p); p += align256(cells * 4);
  uint64_t* off = reinterpret_cast<uint64_t*>(p); p += align256((cells + 1) * 8);
  void* scan_tmp = p;
  const Tuning tn;  // the scan's defaults
  const uint64_t* src = in;
  for (int pass = 0; pass < 4; pass++) {  // bits 32..63; destinations alternate pong, out, pong, out
    uint64_t* dst = (pass & 1) ? out : pong;
    const int shift = 32 + 8 * pass;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, st, src, n, shift, bh, nb);
    launch_scan_u32(bh, cells, off, scan_tmp, tn, st);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(64), 0, st, src, dst, n, shift, off, nb);
    src = dst;
  }
}

}  // namespace sc
