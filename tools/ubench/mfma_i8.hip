// micro-benchmark: what v_mfma_i32_32x32x32_i8 sustains on this chip as a function of (waves per SIMD, independent accumulator
// chains per wave, chain length before the accumulators are re-seeded and read, operand data).  One workgroup = 4 waves (one
// per SIMD), grid = 256 CUs x waves per SIMD: every wave is resident, nothing but MFMAs in the loop.
//   MODE 0: CH chains, each a run of LEN dependent MFMAs, accumulators carried across iterations (the classic GEMM stream)
//   MODE 1: the matcher's stream: per "tile" the accumulators are re-seeded (v_mov from a VGPR), LEN = 4 dependent MFMAs per chain,
//           then 8 v_max3 per chain read them (the fast path of the nearest-neighbour sweep)
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/mfma_i8.hip -o tools/ubench/mfma_i8.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int CH, int MODE>
__global__ __launch_bounds__(256, (CH == 4 && MODE == 1) ? 3 : 5) void k(const v4i *__restrict__ ops, int *__restrict__ out, int iters) {
  v4i a[4], b[CH][4];
  for (int i = 0; i < 4; i++) a[i] = ops[(threadIdx.x & 63) * 4 + i];
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 4; i++) b[c][i] = ops[256 + ((threadIdx.x + 17 * c) & 63) * 4 + i];
  v16i acc[CH];
  for (int c = 0; c < CH; c++)
    for (int r = 0; r < 16; r++) acc[c][r] = r + c;
  int m = 0;
  for (int it = 0; it < iters; it++) {
    if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < CH; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = a[0][r & 3] + r;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
    if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        int x = max(max(acc[c][0], acc[c][1]), acc[c][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) x = max(max(x, acc[c][r]), acc[c][r + 1]);
        m = max(m, max(x, acc[c][15]));
      }
    }
  }
  int s = m;
  for (int c = 0; c < CH; c++)
    for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, int MODE>
static void run(const v4i *ops, int *out, int wps, const char *data) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wps;
  hipLaunchKernelGGL((k<CH, MODE>), dim3(grid), dim3(256), 0, 0, ops, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<CH, MODE>), dim3(grid), dim3(256), 0, 0, ops, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double nm = (double)grid * 4 * iters * 4 * CH;
  printf("mode %d chains %d waves/SIMD %d data %-6s: %.3f ms, %.2f POP/s, %.1f ns per MFMA and SIMD (32 cycles = %.2f GHz at full rate)\n", MODE, CH, wps, data, ms,
         nm * 65536 / (ms * 1e-3) / 1e15, ms * 1e6 / (nm / 1024), 32.0 / (ms * 1e6 / (nm / 1024)));
}

int main() {
  v4i *ops; int *out;
  hipMalloc(&ops, 2 * 256 * 16); hipMalloc(&out, 4 * 256 * 256 * 8);
  for (int data = 0; data < 2; data++) {
    std::vector<int> h(2 * 256 * 4);
    for (auto &x : h) x = data ? (int)(((unsigned)rand() << 16) ^ (unsigned)rand()) : 0;
    hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char *dn = data ? "random" : "zero";
    for (int wps : {1, 2, 3, 4, 5}) {
      run<1, 0>(ops, out, wps, dn);
      run<2, 0>(ops, out, wps, dn);
      run<4, 0>(ops, out, wps, dn);
      run<2, 1>(ops, out, wps, dn);
      if (wps <= 3) run<4, 1>(ops, out, wps, dn);
    }
  }
  return 0;
}
