// Micro-benchmark: a chain of N small dependent kernels launched (a) one by one on a stream, (b) as a captured hipGraph.
// Question: does a graph shrink the gap between dependent launches (the pyramid of the small octaves is ~40 launches of
// 6-10 us each with ~5 us between them)?   hipcc --offload-arch=gfx950 -O3 graph_gap.hip -o graph_gap && ./graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void step(float *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
  const int n = 1 << 18, N = 40, reps = 50;
  float *d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemset(d, 0, n * sizeof(float)));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto chain = [&]() { for (int k = 0; k < N; k++) hipLaunchKernelGGL(step, dim3(n / 256), dim3(256), 0, s, d, n); };
  for (int w = 0; w < 3; w++) chain();
  CK(hipStreamSynchronize(s));
  float ms = 0, best = 1e9;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("stream launches : %.1f us per chain of %d = %.2f us per kernel\n", best * 1e3, N, best * 1e3 / N);
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(); CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  best = 1e9;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("graph launch    : %.1f us per chain of %d = %.2f us per kernel\n", best * 1e3, N, best * 1e3 / N);
  // one kernel alone, for the kernel's own duration
  best = 1e9;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(step, dim3(n / 256), dim3(256), 0, s, d, n); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("one kernel      : %.2f us\n", best * 1e3);
  return 0;
}
