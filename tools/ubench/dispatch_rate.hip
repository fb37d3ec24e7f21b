// How fast does the chip get through workgroups that have nothing to do?  (r04b: the Gram filter's launch holds hundreds of
// workgroups that exit at once — splits its near rows do not use.)  512 threads, 56 KiB of LDS, 128 VGPRs per lane: the
// shape of score_gram_kernel.   hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_rate.hip -o /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void shell(const unsigned* __restrict__ flag, unsigned* out, unsigned work) {
  __shared__ unsigned lds[14336];
  if (blockIdx.x >= work) { if (flag[0] == 12345u) out[blockIdx.x] = 1; return; }
  // a working block: ~10 us of dependent LDS traffic
  unsigned v = threadIdx.x;
  for (int i = 0; i < 2000; i++) { lds[(v + i) % 14336] = v; __syncthreads(); v = lds[(v * 7 + i) % 14336] + i; }
  float f[100];
  for (int i = 0; i < 100; i++) f[i] = v * i;
  for (int i = 0; i < 100; i++) v += (unsigned)f[(i * 7) % 100];
  out[blockIdx.x] = v;
}
int main() {
  unsigned *flag, *out;
  hipMalloc(&flag, 4); hipMemset(flag, 0, 4); hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grids[] = {1, 256, 512, 1024, 2048, 4096, 8192};
  for (unsigned work : {0u, 256u}) for (unsigned g : grids) {
    if (g < work) continue;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(shell, dim3(g), dim3(512), 0, 0, flag, out, work);
    hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 20; r++) hipLaunchKernelGGL(shell, dim3(g), dim3(512), 0, 0, flag, out, work); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("working %u of %u workgroups: %.2f us per launch\n", work, g, ms * 1000 / 20);
  }
  return 0;
}
