// VALU issue rate on gfx950: cycles per wave64 instruction and SIMD for a few instruction classes.
// Each wave runs ITER iterations of 32 independent instructions of one class (no memory traffic); the grid puts W waves on
// every SIMD.  Prints shader cycles per instruction and SIMD from s_memtime of wave 0 and from wall time.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int ITER = 32768;
#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float a, float b, int ia) {
  float v[32];
  int iv[32];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[16]; const f2 pa = {a, b};
  double dv[16]; const double da = a;
  const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)ia;
#pragma unroll
  for (int i = 0; i < 16; i++) { pk[i] = f2{threadIdx.x * 0.001f + i, 1.f}; dv[i] = threadIdx.x + i; }
#pragma unroll
  for (int i = 0; i < 32; i++) { v[i] = threadIdx.x * 0.001f + i; iv[i] = threadIdx.x + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
      if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
      if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
      if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(a));
      if (KIND == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if (KIND == 7) asm volatile("v_floor_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 8) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(iv[i]) : "v"(v[i]));
      if (KIND == 9) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iv[i]));
      if (KIND == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(iv[i]) : "v"(ia));
      if (KIND == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc");
      if (KIND == 20) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
      if (KIND == 21) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a) : "s20", "s21");
      if (KIND == 22) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_add_f32 %3, %3, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]), "+v"(v[(i + 1) & 31]), "+v"(v[(i + 2) & 31]) : "v"(a) : "vcc");
      if (KIND == 23) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a), "s"(mask) : "vcc");
      if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(mask));
      if (KIND == 13 && i < 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(pa));
      if (KIND == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(iv[i]) : "v"(ia));
      if (KIND == 15) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if (KIND == 16) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if (KIND == 17) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[i]) : "v"(iv[i]));
      if (KIND == 18) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dv[i & 15]) : "v"(da));
      if (KIND == 19) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dv[i & 15]) : "v"(da));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; int si = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) { s += v[i]; si += iv[i]; }
#pragma unroll
  for (int i = 0; i < 16; i++) s += pk[i].x + pk[i].y + (float)dv[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + si;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(const char *name, int waves_per_simd, float *out, unsigned long long *cyc) {
  const int blocks = 256 * waves_per_simd;   // 4 waves per block = one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, cyc, 1.0001f, 0.5f, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(out, cyc, 1.0001f, 0.5f, 3);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double inst = (double)ITER * 32 * waves_per_simd;     // per SIMD
  printf("%-26s %d waves/SIMD: %.2f cycles per instruction and SIMD by s_memtime (100 MHz ticks -> x24 @2.4GHz: %.2f), %.2f by wall @2.4 GHz\n", name, waves_per_simd,
         (double)c / inst, (double)c * 24.0 / inst, ms * 1e-3 * 2.4e9 / inst);
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * 256 * 256 * 8); hipMalloc(&cyc, 8);
  for (int w : {4, 8}) {
    run<20>("cmp+cnd vcc (pair)", w, out, cyc); run<21>("cmp+cnd sgpr (pair)", w, out, cyc); run<22>("cmp,add,add,cnd vcc (4)", w, out, cyc);
    run<23>("s_mov vcc + cnd (pair)", w, out, cyc); run<3>("cnd vcc const", w, out, cyc); run<12>("cnd sgpr const", w, out, cyc);
  }
  return 0;
}
