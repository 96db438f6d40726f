// micro-benchmark: cost of an 8-byte gather per wave instruction and CU as a function of how the 64 lanes' addresses fall
// into cache lines.  The image (8 MB) stays L2 / MALL resident.  pattern: lane l of wave-instruction i reads pixel
//   base(wave, i) + f(l):  0: l * 1 px (coalesced)  1: l * 2 px  2: (l & 3) * 1 + (l >> 2) * 64 px (quads share a line)
//   3: l * 64 px (every lane its own line, one row)  4: l * 1920 px (every lane its own row)  5: (l&7)*1 + (l>>3)*1920 (8 x 8 block)
//   6: the 8 x 8 block of 5, but the 8 loads of an iteration walk 8 px to the right each (the same rows: L1 reuse, as a row of tiles does)
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((packed, aligned(4))) PixPair { float a, b; };
template <int PAT>
__global__ __launch_bounds__(256) void k(const float *img, float *out, int iters, int npx) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * 4 + (threadIdx.x >> 6));
  int off;
  if (PAT == 0) off = lane; else if (PAT == 1) off = lane * 2; else if (PAT == 2) off = (lane & 3) + (lane >> 2) * 64;
  else if (PAT == 3) off = lane * 64; else if (PAT == 4) off = lane * 1920; else off = (lane & 7) + (lane >> 3) * 1920;
  unsigned walk = 0;
  float acc = 0;
  unsigned base = (unsigned)gw * 7919u;
  for (int it = 0; it < iters; it++) {
    PixPair v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (PAT != 6 || q == 0) { base = base * 1664525u + 1013904223u; walk = 0; } else walk += 8;
      const unsigned p = ((base >> 8) % (unsigned)(npx - 64 * 1920 - 80)) + off + walk;
      v[q] = *(const PixPair *)(img + p);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) acc += v[q].a + v[q].b;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int npx = 1920 * 1080;
  float *img, *out;
  hipMalloc(&img, npx * 4); hipMalloc(&out, 4 * 256 * 4096); hipMemset(img, 0, npx * 4);
  const int iters = 400;
  for (int pat = 0; pat < 7; pat++)
    for (int bpc : {2, 4}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int grid = 256 * bpc;
      auto launch = [&] {
        switch (pat) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
          default: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, img, out, iters, npx); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double ops_per_cu = (double)iters * 8 * 4 * bpc;
      printf("pattern %d  %2d waves/CU: %.3f ms, %.1f ns = %.0f cycles (2.1 GHz) per wave-load per CU\n", pat, 4 * bpc, ms, ms * 1e6 / ops_per_cu, ms * 1e6 / ops_per_cu * 2.1);
    }
  return 0;
}
