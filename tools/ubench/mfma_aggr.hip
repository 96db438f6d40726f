// aggressor library for tools/stress_spin.py (AGGR_MFMA=<mode>): nothing but MFMA streams, to find what in the matcher's pass 1
// disturbs the fp64 work of co-resident waves of other streams (DESIGN.md "The matcher and its neighbours").
//   mode = dtype * 100 + chains * 10 + kind;   dtype 0: i32_32x32x32_i8, 1: f32_32x32x16_bf16, 2: i32_16x16x64_i8
//   chains: independent accumulators per wave (1, 2, 4);  kind 0: accumulators carried, 1: re-seeded every 4 MFMAs and read by VALU max
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/mfma_aggr.hip -o tools/ubench/libmfmaaggr.so
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));

template <int CH, int KIND, int DT>
__global__ __launch_bounds__(256) void k(const v4i *__restrict__ ops, int *__restrict__ out, int iters) {
  v4i a[4], b[CH][4];
  for (int i = 0; i < 4; i++) a[i] = ops[(threadIdx.x & 63) * 4 + i];
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 4; i++) b[c][i] = ops[256 + ((threadIdx.x + 17 * c) & 63) * 4 + i];
  int s = 0;
  if (KIND == 9 || KIND == 10) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");   // 512 registers: nothing else fits on the SIMD
  if (KIND == 11) asm volatile("v_mov_b32 v255, 0" ::: "v255");                                                       // 256 registers: 2 waves per SIMD
  if (DT == 0) {
    v16i acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) acc[c][r] = r + c;
    int m = 0;
    for (int it = 0; it < iters; it++) {
      if (KIND == 1 || KIND >= 9) {
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[c][r] = a[0][r & 3] + r;
      }
      if (KIND <= 1 || KIND >= 9) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
      } else if (KIND == 2) {          // chain-major: the four dependent MFMAs of a chain back to back, then the next chain
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
          for (int ks = 0; ks < 4; ks++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
      } else if (KIND >= 3 && KIND <= 6) {   // interleaved chains with s_nop (0, 1, 3, 7) after every MFMA
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int c = 0; c < CH; c++) {
            acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
            if (KIND == 3) asm volatile("s_nop 0"); else if (KIND == 4) asm volatile("s_nop 1"); else if (KIND == 5) asm volatile("s_nop 3"); else asm volatile("s_nop 7");
          }
      } else if (KIND == 7) {          // interleaved chains with one VALU instruction after every MFMA
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int c = 0; c < CH; c++) {
            acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
            asm volatile("v_add_u32 %0, %0, 1" : "+v"(m));
          }
      } else if (KIND == 8) {          // pairs: c0 c0 c1 c1 c0 c0 c1 c1
#pragma unroll
        for (int kp = 0; kp < 2; kp++)
#pragma unroll
          for (int c = 0; c < CH; c++)
#pragma unroll
            for (int ks = 2 * kp; ks < 2 * kp + 2; ks++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
      }
      if (KIND == 1 || KIND >= 9) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          int x = max(max(acc[c][0], acc[c][1]), acc[c][2]);
#pragma unroll
          for (int r = 3; r < 15; r += 2) x = max(max(x, acc[c][r]), acc[c][r + 1]);
          m = max(m, max(x, acc[c][15]));
        }
      }
    }
    s = m;
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
  } else if (DT == 1) {
    v16f acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) acc[c][r] = r + c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int c = 0; c < CH; c++) {
          v8s aa, bb;
          for (int e = 0; e < 4; e++) { aa[2 * e] = (short)a[ks][e]; aa[2 * e + 1] = (short)(a[ks][e] >> 16); bb[2 * e] = (short)b[c][ks][e]; bb[2 * e + 1] = (short)(b[c][ks][e] >> 16); }
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, aa), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bb), acc[c], 0, 0, 0);
        }
    }
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) s += (int)acc[c][r];
  } else {
    v4i acc[CH * 4];
    for (int c = 0; c < CH * 4; c++) for (int r = 0; r < 4; r++) acc[c][r] = r + c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int c = 0; c < CH * 4; c++) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ks], b[c & (CH - 1)][ks], acc[c], 0, 0, 0);
    }
    for (int c = 0; c < CH * 4; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static v4i *g_ops = nullptr; static int *g_out = nullptr; static hipStream_t g_st;
extern "C" int aggr_launch(int mode, int random, int grid, int iters) {
  if (!g_ops) {
    hipMalloc(&g_ops, 2 * 256 * 16); hipMalloc(&g_out, 4 * 256 * 4096);
    std::vector<int> h(2 * 256 * 4);
    for (auto &x : h) x = random ? (int)(((unsigned)rand() << 16) ^ (unsigned)rand()) & (random == 2 ? 0x3f803f80 : -1) : 0;
    hipMemcpy(g_ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking);
  }
  if (grid > 4096) grid = 4096;
#define L(CH, KIND, DT) hipLaunchKernelGGL((k<CH, KIND, DT>), dim3(grid), dim3(256), 0, g_st, g_ops, g_out, iters)
  switch (mode) {
    case 10: L(1, 0, 0); break; case 20: L(2, 0, 0); break; case 40: L(4, 0, 0); break;
    case 11: L(1, 1, 0); break; case 21: L(2, 1, 0); break; case 41: L(4, 1, 0); break;
    case 22: L(2, 2, 0); break; case 42: L(4, 2, 0); break;
    case 23: L(2, 3, 0); break; case 24: L(2, 4, 0); break; case 25: L(2, 5, 0); break; case 26: L(2, 6, 0); break;
    case 43: L(4, 3, 0); break; case 44: L(4, 4, 0); break; case 45: L(4, 5, 0); break; case 46: L(4, 6, 0); break;
    case 49: L(4, 9, 0); break; case 411: L(4, 11, 0); break; case 29: L(2, 9, 0); break;
    case 27: L(2, 7, 0); break; case 47: L(4, 7, 0); break; case 28: L(2, 8, 0); break; case 48: L(4, 8, 0); break;
    case 110: L(1, 0, 1); break; case 120: L(2, 0, 1); break; case 140: L(4, 0, 1); break;
    case 210: L(1, 0, 2); break; case 220: L(2, 0, 2); break; case 240: L(4, 0, 2); break;
    default: return -1;
  }
  return hipStreamSynchronize(g_st) == hipSuccess ? 0 : 1;
}
