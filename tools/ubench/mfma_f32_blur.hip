// micro-benchmark for VERDICT r05 item 1d: could the row pass of the extraction blur run on the matrix pipe?
//   (1) is v_mfma_f32_32x32x2_f32 chained over K bitwise equal to a left-to-right fmaf chain (what the blur contract needs)?
//   (2) what does a 32 x 32 output tile of a banded (Toeplitz) row pass cost on the matrix pipe against the vector ALU?
// A row pass is out[y][x] = t0 * S[y][x - r] then fma over the taps left to right: as a matrix product out = S_window x T with T the
// (32 + 2r) x 32 band matrix of taps, the contraction index runs over the window columns in ascending order - the same order - with
// zeros outside the band (fma(0, s, acc) = acc exactly for finite s).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_f32_blur.bin mfma_f32_blur.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));

// one wave: D = S (32 x K) x T (K x 32), K even, by K / 2 chained MFMAs.  Operand layout of v_mfma_f32_32x32x2_f32: A: lane l holds
// A[l % 32][l / 32], B: lane l holds B[l / 32][l % 32] (k = l / 32), D register q of lane l = D[(q & 3) + 8 (q >> 2) + 4 (l >> 5)][l & 31]
__global__ __launch_bounds__(64) void mfma_tile(const float *S, const float *T, float *D, int K) {
  const int l = threadIdx.x;
  v16f acc;
  for (int q = 0; q < 16; q++) acc[q] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) {
    const float a = S[(l & 31) * K + k0 + (l >> 5)];
    const float b = T[(k0 + (l >> 5)) * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int q = 0; q < 16; q++) D[((q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[q];
}
// the same product as fmaf chains in ascending k (one thread per output)
__global__ void fma_tile(const float *S, const float *T, float *D, int K) {
  const int i = threadIdx.x / 32 + 8 * blockIdx.x, j = threadIdx.x % 32;
  float s = 0.f;
  for (int k = 0; k < K; k++) s = fmaf(S[i * K + k], T[k * 32 + j], s);
  D[i * 32 + j] = s;
}
// issue rate: REP back-to-back dependent-free MFMAs per wave (4 accumulators), W waves per SIMD
__global__ __launch_bounds__(256) void mfma_rate(float *out, int iters) {
  v16f acc[4];
  for (int u = 0; u < 4; u++) for (int q = 0; q < 16; q++) acc[u][q] = (float)(threadIdx.x + q);
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
  float s = 0.f;
  for (int u = 0; u < 4; u++) for (int q = 0; q < 16; q++) s += acc[u][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void fma_rate(float *out, int iters, float t) {
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = fmaf(t, v[(i + 1) & 31], v[i]);
  float s = 0.f;
  for (int i = 0; i < 32; i++) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  // (1) bitwise check on a banded product: 32 rows, 11 taps (r = 5): K = 42 window columns
  for (int n_tap : {5, 11, 21}) {
    const int r = n_tap / 2, K = 32 + 2 * r;
    std::vector<float> S(32 * K), T(K * 32, 0.f), tap(n_tap);
    srand(7 + n_tap);
    for (auto &v : S) v = (float)(rand() % 65536) / 257.0f;
    double sum = 0;
    for (int j = 0; j < n_tap; j++) { tap[j] = expf(-0.5f * (j - r) * (j - r) / (0.09f * n_tap * n_tap)); sum += tap[j]; }
    for (int j = 0; j < n_tap; j++) tap[j] = (float)(tap[j] / sum);
    for (int x = 0; x < 32; x++) for (int j = 0; j < n_tap; j++) T[(x + j) * 32 + x] = tap[j];     // out column x reads window columns x .. x + 2r
    float *dS, *dT, *dA, *dB;
    hipMalloc(&dS, S.size() * 4); hipMalloc(&dT, T.size() * 4); hipMalloc(&dA, 4096); hipMalloc(&dB, 4096);
    hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_tile, dim3(1), dim3(64), 0, 0, dS, dT, dA, K);
    hipLaunchKernelGGL(fma_tile, dim3(4), dim3(256), 0, 0, dS, dT, dB, K);
    std::vector<float> A(1024), B(1024), C(1024);
    hipMemcpy(A.data(), dA, 4096, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, 4096, hipMemcpyDeviceToHost);
    // the blur's own chain on the host: first tap as a product, then fma left to right
    for (int y = 0; y < 32; y++) for (int x = 0; x < 32; x++) {
      float s = tap[0] * S[y * K + x];
      for (int j = 1; j < n_tap; j++) s = fmaf(tap[j], S[y * K + x + j], s);
      C[y * 32 + x] = s;
    }
    int d_ab = 0, d_ac = 0;
    for (int i = 0; i < 1024; i++) { d_ab += memcmp(&A[i], &B[i], 4) != 0; d_ac += memcmp(&A[i], &C[i], 4) != 0; }
    printf("%2d taps, K = %2d: MFMA chain vs fmaf chain over all K: %d of 1024 outputs differ; vs the blur's row-pass chain: %d differ\n", n_tap, K, d_ab, d_ac);
    hipFree(dS); hipFree(dT); hipFree(dA); hipFree(dB);
  }
  // (2) rates
  float *out; hipMalloc(&out, 4 * 256 * 2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpc : {4, 8}) {                      // waves per CU
    const int grid = 256 * wpc / 4;
    hipLaunchKernelGGL(mfma_rate, dim3(grid), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_rate, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 4 * (wpc / 4.0);          // MFMAs issued per SIMD
    printf("v_mfma_f32_32x32x2_f32, %d waves per CU: %.1f ns per MFMA and SIMD = %.0f cycles at 2.4 GHz (%.1f TFLOP/s)\n", wpc, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4, 2.0 * 32 * 32 * 2 * per_simd * 1024 / (ms * 1e-3) / 1e12);
    hipLaunchKernelGGL(fma_rate, dim3(grid), dim3(256), 0, 0, out, 10, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(fma_rate, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double fper = (double)iters * 32 * (wpc / 4.0);
    printf("v_fma_f32, %d waves per CU: %.2f ns per wave64 FMA and SIMD = %.1f cycles\n", wpc, ms * 1e6 / fper, ms * 1e6 / fper * 2.4);
  }
  printf("a 32 x 32 tile of an n-tap row pass: matrix pipe = (32 + n - 1) / 2 MFMAs, vector ALU = 1024 n / 64 wave FMAs:\n");
  return 0;
}
