// micro-benchmark: cost of fp64 accumulation into LDS bins, per CU, as the SIFT histogram does it
//   mode 0: ds_add_f64 (no return), bank-conflict-free bin layout   mode 1: ds_add_f64, random banks
//   mode 2: read-add-write (ds_read_b64, v_add_f64, ds_write_b64), conflict-free   mode 3: ds_add_f32 conflict-free (reference point)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(const int *bo_in, double *out, int iters, unsigned long long *cyc) {
  extern __shared__ double acc[];   // per wave 1024 doubles
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *a = acc + wv * 1024;
  for (int i = lane; i < 1024; i += 64) a[i] = 0.0;
  __syncthreads();
  const int slot = lane & 31;
  const int halfw = lane >> 5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int idx = lane;
  for (int it = 0; it < iters; it++) {
    int bo[8];
#pragma unroll
    for (int q = 0; q < 8; q++) bo[q] = bo_in[(idx + q * 64) & 4095];
    idx += 512;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const float c = 1.0f + bo[q];
      int d;
      if (MODE == 1) d = halfw * 512 + ((bo[q] * 37 + lane * 13) & 511);          // random banks
      else d = halfw * 512 + (bo[q] & 7) * 32 + slot;                               // own bank pair per lane
      if (MODE == 0 || MODE == 1) __hip_atomic_fetch_add(a + d, (double)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      else if (MODE == 2) a[d] += (double)c;
      else __hip_atomic_fetch_add((float *)(a + d), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  double s = 0;
  for (int i = lane; i < 1024; i += 64) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  int *bo; double *out; unsigned long long *cyc;
  hipMalloc(&bo, 4096 * 4); hipMalloc(&out, 8 * 256 * 2048); hipMalloc(&cyc, 8);
  std::vector<int> h(4096); for (auto &v : h) v = rand() & 7;
  hipMemcpy(bo, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int mode = 0; mode < 4; mode++)
    for (int bpc : {1, 2, 4}) {     // blocks (of 4 waves) per CU
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int grid = 256 * bpc;
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 4 * 1024 * 8, 0, bo, out, iters, cyc);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 4 * 1024 * 8, 0, bo, out, iters, cyc);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 4 * 1024 * 8, 0, bo, out, iters, cyc);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 4 * 1024 * 8, 0, bo, out, iters, cyc);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double ops_per_cu = (double)iters * 8 * 4 * bpc;   // wave-level accumulate instructions per CU
      printf("mode %d  %d waves/CU: %.3f ms, %.1f ns per wave-op per CU  (wave 0: %.1f memtime ticks per op)\n", mode, 4 * bpc, ms, ms * 1e6 / ops_per_cu,
             (double)c / (iters * 8));
    }
  return 0;
}
