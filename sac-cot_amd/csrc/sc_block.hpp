// sc_block.hpp — workgroup-level reduce / exclusive scan helpers (64-lane waves, LDS cross-wave step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

// A result the HOST polls for in pinned memory (sc_capi.hip wait_word): one system-scope release store.
__device__ __forceinline__ void publish_host(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum over the block; lds: >= blockDim.x/64 entries.  Every thread gets the total.
__device__ __forceinline__ uint64_t block_reduce_u64(uint64_t v, uint64_t* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  uint64_t t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += lds[w];
  __syncthreads();
  return t;
}

// exclusive scan of one value per thread across the block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ uint64_t block_exscan_u64(uint64_t v, uint64_t* lds, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint64_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
    uint64_t x = lds[w];
    if (w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// Decoupled look-back (single-pass prefix over the tiles of one launch).  One 64-bit DESCRIPTOR per tile and value:
//   [ epoch : 12 | status : 2 | value : 50 ]     status 0 = nothing yet, 1 = the tile's own sum, 2 = inclusive prefix
// written and read as ONE 8-byte agent-scope atomic (global_store / global_load ... sc1): the value travels inside the
// word that signals it, so no fence and no second hand-off is needed (cdna guide §6 G16 "data-tagged granule").  The
// epoch makes words of earlier launches read as "nothing yet" — no memset between launches (the host zeroes the area
// when the 12-bit epoch wraps or the buffer is new).  Tiles take their index from an atomic ticket, so every predecessor
// of a running tile has started and will publish its sum without waiting for anyone: the look-back cannot deadlock.
// Wave 0 of the tile looks back 64 predecessors per step.  A spin limit turns a protocol failure into a wrong result
// plus an error flag instead of a hang.
// ------------------------------------------------------------------------------------------------
constexpr uint64_t LB_VALUE_MASK = (1ull << 50) - 1ull;
constexpr uint32_t LB_SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ uint64_t lb_pack(uint32_t epoch, uint32_t status, uint64_t v) {
  return ((uint64_t)(epoch & 0xFFFu) << 52) | ((uint64_t)status << 50) | (v & LB_VALUE_MASK);
}
__device__ __forceinline__ void lb_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t lb_load(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by the 64 lanes of ONE wave of tile `tile` (every lane, converged).  NV values per tile (1 or 2), descriptor
// arrays desc[v] of gridDim.x words each; own[v] = this tile's sums.  Returns the exclusive prefixes in pre[v] (the same
// in every lane) and publishes the inclusive ones.  *err is set (lane 0) if the spin limit was hit.
template <int NV>
__device__ __forceinline__ void lb_lookback(uint64_t* const (&desc)[NV], uint32_t tile, uint32_t epoch,
                                            const uint64_t (&own)[NV], uint64_t (&pre)[NV], uint32_t* err) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < NV; v++) pre[v] = 0;
  if (tile == 0) {
    if (lane == 0) {
#pragma unroll
      for (int v = 0; v < NV; v++) lb_store(&desc[v][0], lb_pack(epoch, 2u, own[v]));
    }
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int v = 0; v < NV; v++) lb_store(&desc[v][tile], lb_pack(epoch, 1u, own[v]));
  }
  int64_t hi = (int64_t)tile - 1;  // nearest predecessor not yet accounted for
  uint32_t spins = 0;
  for (;;) {
    const int64_t idx = hi - lane;
    uint64_t d[NV];
    bool valid = true, isp = true;
    uint32_t st0 = 0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      d[v] = idx >= 0 ? lb_load(&desc[v][idx]) : lb_pack(epoch, 2u, 0);  // below tile 0: a virtual prefix of 0
      const uint32_t st = (uint32_t)(d[v] >> 50) & 3u;
      if (v == 0) st0 = st;
      // A tile publishes its NV words one after the other, twice (own sums, then prefixes).  A reader that catches it in
      // between sees words of DIFFERENT status — a prefix in one, an own sum in the other — and must not mix them: the tile
      // counts only once all its words carry the same status (each word's value always matches its own status).
      const bool ok = (uint32_t)(d[v] >> 52) == (epoch & 0xFFFu) && st != 0u && st == st0;
      valid = valid && ok;
      isp = isp && ok && st == 2u;
    }
    const uint64_t pm = __ballot(isp), vm = __ballot(valid);
    const int k = pm ? __builtin_ctzll(pm) : 64;                          // first lane holding an inclusive prefix
    const uint64_t need = k >= 63 ? ~0ull : ((1ull << (k + 1)) - 1ull);   // lanes 0 .. k (all 64 when there is none)
    if ((vm & need) == need) {
#pragma unroll
      for (int v = 0; v < NV; v++) {
        uint64_t x = lane <= k ? (d[v] & LB_VALUE_MASK) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        pre[v] += x;
      }
      if (k < 64) break;
      hi -= 64;
    } else if (++spins > LB_SPIN_LIMIT) {
      if (lane == 0 && err) *err = 1u;
      break;
    } else {
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int v = 0; v < NV; v++) lb_store(&desc[v][tile], lb_pack(epoch, 2u, pre[v] + own[v]));
  }
}

}  // namespace sc
