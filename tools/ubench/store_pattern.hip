// Which store PATTERN of a symmetric-tile kernel reaches the write rate of a plain fill?   bash tools/ubench/run.sh store_pattern
// N = 20 000 (ld = 20 032): the matrix S is 1.6 GB.  Every variant writes every element of S exactly once with 16-byte
// stores (values: a function of the indices, so that nothing is compressed away); only the ORDER and the piece sizes differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
constexpr int N = 20000, LD = 20032, W = LD / 64;  // 313 column blocks

__device__ __forceinline__ void st4(float* p, float v) { *reinterpret_cast<float4*>(p) = make_float4(v, v + 1, v + 2, v + 3); }

// tile (J, h) with TR rows: direct half rows i0..i0+TR-1 x 64 columns of block J (256-byte pieces), mirrored half rows
// J*64..+63 x TR columns (TR*4-byte pieces)
template <int TR>
__device__ __forceinline__ void write_tile(float* S, int J, int h, int lane) {
  const int i0 = h * TR;
  if (i0 >= N) return;
  {  // direct: 16 lanes x 16 B = one 256-byte row piece, 4 rows per instruction
    const int c4 = (lane & 15) * 4, rr = lane >> 4;
    for (int q = 0; q < TR; q += 4) {
      const int r = i0 + q + rr;
      if (r < N) st4(&S[(size_t)r * LD + J * 64 + c4], (float)(r + c4));
    }
  }
  if (h / (64 / TR) == J) return;  // diagonal block: direct half only
  {  // mirrored: row J*64 + cc, columns i0 .. i0 + TR - 1
    constexpr int LP = TR / 4, RPI = 64 / LP;
    const int r4 = (lane % LP) * 4, co = lane / LP;
    for (int c = 0; c < 64; c += RPI) {
      const int row = J * 64 + c + co;
      if (row < N) st4(&S[(size_t)row * LD + i0 + r4], (float)(row + r4));
    }
  }
}
__device__ __forceinline__ void tri_index(int t, int SUB, int& J, int& h) {
  J = (int)((__builtin_sqrtf(8.0f * (float)(t / SUB) + 1.0f) - 1.0f) * 0.5f);
  while (SUB * (J + 1) * (J + 2) / 2 <= t) J++;
  while (SUB * J * (J + 1) / 2 > t) J--;
  h = t - SUB * J * (J + 1) / 2;
}
// V0: the product's order — wave t of a 4-wave workgroup takes tile t (consecutive h of one column block)
__global__ __launch_bounds__(256) void v0_tiles16(float* S, int n_tiles) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tiles) return;
  int J, h; tri_index(t, 4, J, h);
  write_tile<16>(S, J, h, threadIdx.x & 63);
}
// V2: 64-row tiles, one wave each
__global__ __launch_bounds__(64) void v2_tiles64(float* S, int n_tiles) {
  const int t = blockIdx.x;
  if (t >= n_tiles) return;
  int J, h; tri_index(t, 1, J, h);
  write_tile<64>(S, J, h, threadIdx.x & 63);
}
// V3: 16 waves per workgroup = one 64 x 64 block pair handled as 4 x 4 ... here: a workgroup takes 4 consecutive 64-row
// tiles of one column block (256 rows x 256 B direct, 64 rows x 1 KiB mirrored)
__global__ __launch_bounds__(256) void v3_tiles64x4(float* S, int n_tiles) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tiles) return;
  int J, h; tri_index(t, 1, J, h);
  write_tile<64>(S, J, h, threadIdx.x & 63);
}
// V4: row streaming (what a fill does): workgroup b writes 4 KiB contiguous
__global__ __launch_bounds__(256) void v4_rows(float* S, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total4) st4(&S[i * 4], (float)(i & 1023));
}
// V5: V0's tiles, but a workgroup takes the same h of FOUR adjacent column blocks (direct: 16 rows x 1 KiB; mirrored: 64-byte pieces)
__global__ __launch_bounds__(256) void v5_tiles16_across(float* S) {
  const int w = threadIdx.x >> 6;
  // grid: (ceil(W / 4), rows / 16): column group, h
  const int J = blockIdx.x * 4 + w, h = blockIdx.y;
  if (J >= W || h > 4 * J + 3) return;
  write_tile<16>(S, J, h, threadIdx.x & 63);
}
// V6: 8 waves: 2 column blocks x 4 consecutive h (direct 512-byte pieces, mirrored 256-byte pieces)
__global__ __launch_bounds__(512) void v6_tiles16_2x4(float* S) {
  const int w = threadIdx.x >> 6;
  const int J = blockIdx.x * 2 + (w >> 2), h = blockIdx.y * 4 + (w & 3);
  if (J >= W || h > 4 * J + 3) return;
  write_tile<16>(S, J, h, threadIdx.x & 63);
}

int main() {
  float* S; const size_t bytes = (size_t)N * LD * 4;
  hipMalloc(&S, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto bench = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f us  %.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e12);
  };
  const int nt16 = 4 * W * (W + 1) / 2, nt64 = W * (W + 1) / 2;
  bench("plain fill (hipMemsetAsync)", [&] { hipMemsetAsync(S, 0, bytes, 0); });
  bench("V4 row streaming, 16-byte stores", [&] { hipLaunchKernelGGL(v4_rows, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, S, bytes / 16); });
  bench("V0 16-row tiles, 4 consecutive h per workgroup", [&] { hipLaunchKernelGGL(v0_tiles16, dim3((nt16 + 3) / 4), dim3(256), 0, 0, S, nt16); });
  bench("V2 64-row tiles, one wave per workgroup", [&] { hipLaunchKernelGGL(v2_tiles64, dim3(nt64), dim3(64), 0, 0, S, nt64); });
  bench("V3 64-row tiles, 4 consecutive h per workgroup", [&] { hipLaunchKernelGGL(v3_tiles64x4, dim3((nt64 + 3) / 4), dim3(256), 0, 0, S, nt64); });
  bench("V5 16-row tiles, 4 adjacent column blocks", [&] { hipLaunchKernelGGL(v5_tiles16_across, dim3((W + 3) / 4, 4 * W), dim3(256), 0, 0, S); });
  bench("V6 16-row tiles, 2 column blocks x 4 h", [&] { hipLaunchKernelGGL(v6_tiles16_2x4, dim3((W + 1) / 2, W), dim3(512), 0, 0, S); });
  return 0;
}
