// ubench_gridsync.hip — what a grid-wide barrier costs on MI355X (8 XCDs) next to a kernel-launch boundary (~4.6 us
// per dependent launch, measured): a cooperative kernel doing K cooperative_groups::grid().sync() rounds.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gridsync.hip -o tools/ubench_gridsync && tools/ubench_gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_sync(unsigned* out, int rounds) {
  cg::grid_group g = cg::this_grid();
  unsigned acc = 0;
  for (int r = 0; r < rounds; r++) {
    acc += threadIdx.x ^ (unsigned)r;
    g.sync();
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_empty(unsigned* out) {
  if (threadIdx.x == 1234567) out[0] = 1;
}

int main() {
  unsigned* out;
  hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 1024, 2048}) {
    for (int rounds : {0, 10, 100}) {
      void* args[] = {&out, &rounds};
      hipError_t err = hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(256), args, 0, 0);
      if (err != hipSuccess) { printf("blocks %d: cooperative launch refused (%s)\n", blocks, hipGetErrorString(err)); break; }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < 20; it++) hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(256), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d  syncs %3d : %8.2f us per launch\n", blocks, rounds, ms * 1000.f / 20);
    }
  }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 200; it++) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("back-to-back empty launches (1024 blocks): %.2f us each\n", ms * 1000.f / 200);
  return 0;
}
