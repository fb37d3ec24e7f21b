// sc_sort.hip — ranked order of the T selected triangles: ascending u64 radix sort of
// (~key << 32 | position) == (key descending, ordinal ascending).  T <= a few 1e5 keys; rocPRIM's device
// radix sort (ROCm-native, header-only) is used for this one non-hot step (SURVEY.md §2 allows it).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "sc_kernels.hpp"

namespace sc {

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  if (n == 0) return 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, 0, false);
  return bytes;
}

void launch_sort_u64(const uint64_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes, hipStream_t st) {
  if (n == 0) return;
  (void)rocprim::radix_sort_keys(temp, temp_bytes, in, out, n, 0, 64, st, false);
}

}  // namespace sc
