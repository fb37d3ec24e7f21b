// What does ONE more launch in a chain of dependent launches cost on this part, and does a HIP graph change it?
// A frame of the path is 17 dependent launches; an empty launch behind another costs ~3.3 us (dispatch_rate.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_chain.hip -o /tmp/launch_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void link(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned n) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[(i * 7u + 1u) % n] + 1u;   // every launch reads what the one before wrote
}
static float run_stream(hipStream_t st, unsigned* a, unsigned* b, unsigned n, unsigned grid, int chain, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) for (int k = 0; k < chain; k++) hipLaunchKernelGGL(link, dim3(grid), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; r++) for (int k = 0; k < chain; k++) hipLaunchKernelGGL(link, dim3(grid), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / (reps * chain);
}
static float run_graph(hipStream_t st, unsigned* a, unsigned* b, unsigned n, unsigned grid, int chain, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int k = 0; k < chain; k++) hipLaunchKernelGGL(link, dim3(grid), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n);
  hipStreamEndCapture(st, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; r++) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * chain);
}
int main() {
  const unsigned n = 1u << 20;
  unsigned *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (unsigned grid : {1u, 64u, 256u, 2048u})
    for (int chain : {1, 17}) {
      const unsigned nn = grid * 256 < n ? grid * 256 : n;
      const float s = run_stream(st, a, b, nn, grid, chain, 200 / chain + 20);
      const float g = run_graph(st, a, b, nn, grid, chain, 200 / chain + 20);
      printf("%4u workgroups, chain of %2d: stream launches %.2f us per kernel, one graph per chain %.2f us per kernel\n", grid, chain, s, g);
    }
  return 0;
}
