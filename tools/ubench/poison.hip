// Leaves a chosen bit pattern in every VGPR (all 64 lanes) of 128-VGPR waves and / or in 64 KB of LDS per workgroup, then exits:
// what a wave of another kernel that starts on the same SIMD / LDS region afterwards finds in registers and LDS it has not
// written yet.  Used by tools/stress_match.py ("poison" aggressors) to test whether a kernel of the library reads registers,
// inactive lanes or LDS words that it never initialised (round 3, DESIGN.md "The matcher and its neighbours").
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/poison.hip -o tools/ubench/libpoison.so
#include <hip/hip_runtime.h>

template <int MODE>
__global__ __launch_bounds__(256) void poison_kernel(unsigned pat, unsigned *__restrict__ sink) {
  extern __shared__ unsigned lds[];
  if (MODE & 2) {
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = pat;
    __syncthreads();
    if (lds[(threadIdx.x * 37) & 16383] == 0x12345u) sink[0] = 1;
  }
  if (MODE & 1)
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %0\n v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n v_mov_b32 v64, %0\n v_mov_b32 v65, %0\n v_mov_b32 v66, %0\n v_mov_b32 v67, %0\n v_mov_b32 v68, %0\n v_mov_b32 v69, %0\n v_mov_b32 v70, %0\n v_mov_b32 v71, %0\n v_mov_b32 v72, %0\n v_mov_b32 v73, %0\n v_mov_b32 v74, %0\n v_mov_b32 v75, %0\n v_mov_b32 v76, %0\n v_mov_b32 v77, %0\n v_mov_b32 v78, %0\n v_mov_b32 v79, %0\n v_mov_b32 v80, %0\n v_mov_b32 v81, %0\n v_mov_b32 v82, %0\n v_mov_b32 v83, %0\n v_mov_b32 v84, %0\n v_mov_b32 v85, %0\n v_mov_b32 v86, %0\n v_mov_b32 v87, %0\n v_mov_b32 v88, %0\n v_mov_b32 v89, %0\n v_mov_b32 v90, %0\n v_mov_b32 v91, %0\n v_mov_b32 v92, %0\n v_mov_b32 v93, %0\n v_mov_b32 v94, %0\n v_mov_b32 v95, %0\n v_mov_b32 v96, %0\n v_mov_b32 v97, %0\n v_mov_b32 v98, %0\n v_mov_b32 v99, %0\n v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0\n v_mov_b32 v112, %0\n v_mov_b32 v113, %0\n v_mov_b32 v114, %0\n v_mov_b32 v115, %0\n v_mov_b32 v116, %0\n v_mov_b32 v117, %0\n v_mov_b32 v118, %0\n v_mov_b32 v119, %0\n v_mov_b32 v120, %0\n v_mov_b32 v121, %0\n v_mov_b32 v122, %0\n v_mov_b32 v123, %0\n v_mov_b32 v124, %0\n v_mov_b32 v125, %0\n v_mov_b32 v126, %0\n v_mov_b32 v127, %0"
                 :: "s"(pat) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

static hipStream_t g_stream;
extern "C" int poison_launch(int mode, unsigned pattern, int blocks, int sync) {
  static unsigned *sink = nullptr;
  if (!sink) {
    if (hipMalloc(&sink, 64) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  const size_t lds = (mode & 2) ? 65536 : 0;
  if (mode == 1) hipLaunchKernelGGL(poison_kernel<1>, dim3(blocks), dim3(256), lds, g_stream, pattern, sink);
  else if (mode == 2) hipLaunchKernelGGL(poison_kernel<2>, dim3(blocks), dim3(256), lds, g_stream, pattern, sink);
  else hipLaunchKernelGGL(poison_kernel<3>, dim3(blocks), dim3(256), lds, g_stream, pattern, sink);
  if (hipGetLastError() != hipSuccess) return 3;
  if (sync && hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  return 0;
}
