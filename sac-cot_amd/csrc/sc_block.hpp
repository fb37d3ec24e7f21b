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

}  // namespace sc
