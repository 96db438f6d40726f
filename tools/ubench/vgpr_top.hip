// Does a wave that writes the LAST registers of a 128-VGPR allocation disturb the registers of ANOTHER wave on its SIMD?
// (Round 3: match_nn1_kernel, 128 VGPRs = 4 waves per SIMD, with `ds_read_b128 v[124:127]` in its loop, intermittently changed
// single results of kernels of OTHER streams - one keypoint's shape, one descriptor - and stopped doing so in every build that
// left v126 / v127 unused; tools/stress_match.py.)  Victim waves park constants in v0..v7 and spin; aggressor workgroups
// (launch_bounds 256, v124..v127 in the clobber list = 128 VGPRs allocated) run one of several instructions that write v[124:127]
// on a second stream.  The host reports the victim lanes whose constants changed and what they changed to.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/vgpr_top.hip -o /tmp/vgpr_top && /tmp/vgpr_top
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void victim(unsigned *__restrict__ out, int iters) {
  unsigned r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "v_mov_b32 v0, 0x5a5a0000\n v_mov_b32 v1, 0x5a5a0001\n v_mov_b32 v2, 0x5a5a0002\n v_mov_b32 v3, 0x5a5a0003\n"
      "v_mov_b32 v4, 0x5a5a0004\n v_mov_b32 v5, 0x5a5a0005\n v_mov_b32 v6, 0x5a5a0006\n v_mov_b32 v7, 0x5a5a0007\n"
      "s_mov_b32 s8, %8\n"
      "1:\n s_sleep 1\n s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
      "v_mov_b32 %0, v0\n v_mov_b32 %1, v1\n v_mov_b32 %2, v2\n v_mov_b32 %3, v3\n"
      "v_mov_b32 %4, v4\n v_mov_b32 %5, v5\n v_mov_b32 %6, v6\n v_mov_b32 %7, v7\n"
      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
      : "s"(iters)
      : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "s8", "scc");
  unsigned *o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 8;
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; o[6] = r6; o[7] = r7;
}

// MODE 0: ds_read_b128 v[124:127]   1: ds_read_b128 v[120:123] (control: same allocation, the top four untouched)
//      2: v_mov_b32 v124..v127      3: global_load_dwordx4 v[124:127]   4: ds_read_b128 v[122:125] (v126, v127 untouched)
template <int MODE>
__global__ __launch_bounds__(256) void aggressor(unsigned *__restrict__ sink, const uint4 *__restrict__ src, int iters) {
  __shared__ uint4 lds[256];
  lds[threadIdx.x] = make_uint4(0xdead0000u | threadIdx.x, 0xdead1000u | threadIdx.x, 0xdead2000u | threadIdx.x, 0xdead3000u | threadIdx.x);
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds + threadIdx.x);     // LDS byte address
  const uint4 *gp = src + threadIdx.x;
  unsigned acc = 0;
  if (MODE == 0)
    asm volatile("s_mov_b32 s8, %2\n 1:\n ds_read_b128 v[124:127], %1\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %0, %0, v124\n v_xor_b32 %0, %0, v127\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(addr), "s"(iters) : "v124", "v125", "v126", "v127", "s8", "scc", "memory");
  else if (MODE == 1)
    asm volatile("s_mov_b32 s8, %2\n 1:\n ds_read_b128 v[120:123], %1\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %0, %0, v120\n v_xor_b32 %0, %0, v123\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(addr), "s"(iters) : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s8", "scc", "memory");
  else if (MODE == 2)
    asm volatile("s_mov_b32 s8, %2\n 1:\n v_mov_b32 v124, %1\n v_mov_b32 v125, %1\n v_mov_b32 v126, %1\n v_mov_b32 v127, %1\n v_xor_b32 %0, %0, v124\n v_xor_b32 %0, %0, v127\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(addr | 0xbeef0000u), "s"(iters) : "v124", "v125", "v126", "v127", "s8", "scc", "memory");
  else if (MODE == 3)
    asm volatile("s_mov_b32 s8, %2\n 1:\n global_load_dwordx4 v[124:127], %1, off\n s_waitcnt vmcnt(0)\n v_xor_b32 %0, %0, v124\n v_xor_b32 %0, %0, v127\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(gp), "s"(iters) : "v124", "v125", "v126", "v127", "s8", "scc", "memory");
  else if (MODE == 5)     // the matcher's pair: the LDS read into the top four registers feeds an MFMA as its A operand
    asm volatile("s_mov_b32 s8, %2\n 1:\n ds_read_b128 v[124:127], %1\n s_waitcnt lgkmcnt(0)\n"
                 "v_mfma_i32_32x32x32_i8 v[0:15], v[124:127], v[16:19], v[0:15]\n v_mfma_i32_32x32x32_i8 v[20:35], v[124:127], v[16:19], v[20:35]\n s_nop 7\n s_nop 7\n"
                 "v_xor_b32 %0, %0, v0\n v_xor_b32 %0, %0, v35\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(addr), "s"(iters) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",
                   "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v124", "v125", "v126",
                   "v127", "s8", "scc", "memory");
  else
    asm volatile("s_mov_b32 s8, %2\n 1:\n ds_read_b128 v[122:125], %1\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %0, %0, v122\n v_xor_b32 %0, %0, v125\n"
                 "s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
                 : "+v"(acc) : "v"(addr), "s"(iters) : "v122", "v123", "v124", "v125", "v126", "v127", "s8", "scc", "memory");
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int MODE>
static int run(const char *what, hipStream_t sa, hipStream_t sb, unsigned *out, unsigned *sink, const uint4 *src, int rounds) {
  const int vblocks = 2048;
  std::vector<unsigned> h((size_t)vblocks * 64 * 8);
  long bad_lanes = 0, total = 0;
  unsigned sample[4] = {0, 0, 0, 0};
  int reg_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ns = 0;
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL(victim, dim3(vblocks), dim3(64), 0, sa, out, 4000);
    for (int q = 0; q < 6; q++) hipLaunchKernelGGL(aggressor<MODE>, dim3(4096), dim3(256), 0, sb, sink, src, 64);
    CHECK(hipStreamSynchronize(sa));
    CHECK(hipStreamSynchronize(sb));
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i += 8) {
      bool bad = false;
      for (int j = 0; j < 8; j++)
        if (h[i + j] != (0x5a5a0000u | j)) { bad = true; reg_hist[j]++; if (ns < 4) sample[ns++] = h[i + j]; }
      bad_lanes += bad; total++;
    }
  }
  printf("%-44s %ld of %ld victim lanes changed; per register v0..v7: %d %d %d %d %d %d %d %d; values seen: %08x %08x %08x %08x\n", what, bad_lanes, total,
         reg_hist[0], reg_hist[1], reg_hist[2], reg_hist[3], reg_hist[4], reg_hist[5], reg_hist[6], reg_hist[7], sample[0], sample[1], sample[2], sample[3]);
  return 0;
}

// aggressor launcher for tools/stress_match.py ("top<mode>" aggressors; build with -shared -fPIC -DVGPR_TOP_LIB)
extern "C" int top_launch(int mode, int blocks, int iters) {
  static hipStream_t st = nullptr;
  static unsigned *sink = nullptr; static uint4 *src = nullptr;
  if (!st) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc(&sink, 65536 * 4) != hipSuccess || hipMalloc(&src, 256 * 16) != hipSuccess) return 1;
    if (hipMemset(src, 0x77, 256 * 16) != hipSuccess) return 1;
  }
  switch (mode) {
    case 0: hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
    case 1: hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
    case 2: hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
    case 3: hipLaunchKernelGGL(aggressor<3>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
    case 4: hipLaunchKernelGGL(aggressor<4>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
    default: hipLaunchKernelGGL(aggressor<5>, dim3(blocks), dim3(256), 0, st, sink, src, iters); break;
  }
  if (hipGetLastError() != hipSuccess) return 2;
  return hipStreamSynchronize(st) == hipSuccess ? 0 : 3;
}

#ifndef VGPR_TOP_LIB
int main() {
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned *out, *sink; uint4 *src;
  CHECK(hipMalloc(&out, (size_t)2048 * 64 * 8 * 4));
  CHECK(hipMalloc(&sink, 4096 * 4));
  CHECK(hipMalloc(&src, 256 * 16));
  CHECK(hipMemset(src, 0x77, 256 * 16));
  const int rounds = 40;
  if (run<1>("control: ds_read_b128 v[120:123]", sa, sb, out, sink, src, rounds)) return 1;
  if (run<4>("ds_read_b128 v[122:125]", sa, sb, out, sink, src, rounds)) return 1;
  if (run<0>("ds_read_b128 v[124:127]", sa, sb, out, sink, src, rounds)) return 1;
  if (run<2>("v_mov_b32 v124..v127", sa, sb, out, sink, src, rounds)) return 1;
  if (run<3>("global_load_dwordx4 v[124:127]", sa, sb, out, sink, src, rounds)) return 1;
  if (run<5>("ds_read_b128 v[124:127] -> 2 x v_mfma srcA", sa, sb, out, sink, src, rounds)) return 1;
  return 0;
}
#endif
