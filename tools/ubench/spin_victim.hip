// A wave that only parks constants in v0..v95 (and a pattern in 16 KB of LDS), sleeps, and counts how many of them changed:
// run next to another context's match_nn1_kernel (tools/stress_match.py, "spin" victim) it tells whether that kernel's
// neighbours lose REGISTER or LDS contents - with no code of the library on the victim's side.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/spin_victim.hip -o tools/ubench/libspin.so
#include <hip/hip_runtime.h>
#include <cstdlib>

__global__ __launch_bounds__(64) void spin_kernel(int iters, unsigned *__restrict__ counts) {
  __shared__ unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xa5000000u | i;
  __syncthreads();
  unsigned bad = 0;
  asm volatile("v_mov_b32 v0, 0x5a000000\n v_mov_b32 v1, 0x5a000001\n v_mov_b32 v2, 0x5a000002\n v_mov_b32 v3, 0x5a000003\n v_mov_b32 v4, 0x5a000004\n v_mov_b32 v5, 0x5a000005\n v_mov_b32 v6, 0x5a000006\n v_mov_b32 v7, 0x5a000007\n v_mov_b32 v8, 0x5a000008\n v_mov_b32 v9, 0x5a000009\n v_mov_b32 v10, 0x5a00000a\n v_mov_b32 v11, 0x5a00000b\n v_mov_b32 v12, 0x5a00000c\n v_mov_b32 v13, 0x5a00000d\n v_mov_b32 v14, 0x5a00000e\n v_mov_b32 v15, 0x5a00000f\n v_mov_b32 v16, 0x5a000010\n v_mov_b32 v17, 0x5a000011\n v_mov_b32 v18, 0x5a000012\n v_mov_b32 v19, 0x5a000013\n v_mov_b32 v20, 0x5a000014\n v_mov_b32 v21, 0x5a000015\n v_mov_b32 v22, 0x5a000016\n v_mov_b32 v23, 0x5a000017\n v_mov_b32 v24, 0x5a000018\n v_mov_b32 v25, 0x5a000019\n v_mov_b32 v26, 0x5a00001a\n v_mov_b32 v27, 0x5a00001b\n v_mov_b32 v28, 0x5a00001c\n v_mov_b32 v29, 0x5a00001d\n v_mov_b32 v30, 0x5a00001e\n v_mov_b32 v31, 0x5a00001f\n v_mov_b32 v32, 0x5a000020\n v_mov_b32 v33, 0x5a000021\n v_mov_b32 v34, 0x5a000022\n v_mov_b32 v35, 0x5a000023\n v_mov_b32 v36, 0x5a000024\n v_mov_b32 v37, 0x5a000025\n v_mov_b32 v38, 0x5a000026\n v_mov_b32 v39, 0x5a000027\n v_mov_b32 v40, 0x5a000028\n v_mov_b32 v41, 0x5a000029\n v_mov_b32 v42, 0x5a00002a\n v_mov_b32 v43, 0x5a00002b\n v_mov_b32 v44, 0x5a00002c\n v_mov_b32 v45, 0x5a00002d\n v_mov_b32 v46, 0x5a00002e\n v_mov_b32 v47, 0x5a00002f\n v_mov_b32 v48, 0x5a000030\n v_mov_b32 v49, 0x5a000031\n v_mov_b32 v50, 0x5a000032\n v_mov_b32 v51, 0x5a000033\n v_mov_b32 v52, 0x5a000034\n v_mov_b32 v53, 0x5a000035\n v_mov_b32 v54, 0x5a000036\n v_mov_b32 v55, 0x5a000037\n v_mov_b32 v56, 0x5a000038\n v_mov_b32 v57, 0x5a000039\n v_mov_b32 v58, 0x5a00003a\n v_mov_b32 v59, 0x5a00003b\n v_mov_b32 v60, 0x5a00003c\n v_mov_b32 v61, 0x5a00003d\n v_mov_b32 v62, 0x5a00003e\n v_mov_b32 v63, 0x5a00003f\n v_mov_b32 v64, 0x5a000040\n v_mov_b32 v65, 0x5a000041\n v_mov_b32 v66, 0x5a000042\n v_mov_b32 v67, 0x5a000043\n v_mov_b32 v68, 0x5a000044\n v_mov_b32 v69, 0x5a000045\n v_mov_b32 v70, 0x5a000046\n v_mov_b32 v71, 0x5a000047\n v_mov_b32 v72, 0x5a000048\n v_mov_b32 v73, 0x5a000049\n v_mov_b32 v74, 0x5a00004a\n v_mov_b32 v75, 0x5a00004b\n v_mov_b32 v76, 0x5a00004c\n v_mov_b32 v77, 0x5a00004d\n v_mov_b32 v78, 0x5a00004e\n v_mov_b32 v79, 0x5a00004f\n v_mov_b32 v80, 0x5a000050\n v_mov_b32 v81, 0x5a000051\n v_mov_b32 v82, 0x5a000052\n v_mov_b32 v83, 0x5a000053\n v_mov_b32 v84, 0x5a000054\n v_mov_b32 v85, 0x5a000055\n v_mov_b32 v86, 0x5a000056\n v_mov_b32 v87, 0x5a000057\n v_mov_b32 v88, 0x5a000058\n v_mov_b32 v89, 0x5a000059\n v_mov_b32 v90, 0x5a00005a\n v_mov_b32 v91, 0x5a00005b\n v_mov_b32 v92, 0x5a00005c\n v_mov_b32 v93, 0x5a00005d\n v_mov_b32 v94, 0x5a00005e\n v_mov_b32 v95, 0x5a00005f\n"
               "s_mov_b32 s8, %1\n 1:\n s_sleep 1\n s_sub_u32 s8, s8, 1\n s_cmp_lg_u32 s8, 0\n s_cbranch_scc1 1b\n"
               "v_cmp_ne_u32_e32 vcc, 0x5a000000, v0\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000001, v1\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000002, v2\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000003, v3\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000004, v4\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000005, v5\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000006, v6\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000007, v7\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000008, v8\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000009, v9\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000a, v10\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000b, v11\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000c, v12\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000d, v13\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000e, v14\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00000f, v15\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000010, v16\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000011, v17\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000012, v18\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000013, v19\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000014, v20\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000015, v21\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000016, v22\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000017, v23\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000018, v24\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000019, v25\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001a, v26\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001b, v27\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001c, v28\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001d, v29\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001e, v30\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00001f, v31\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000020, v32\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000021, v33\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000022, v34\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000023, v35\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000024, v36\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000025, v37\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000026, v38\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000027, v39\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000028, v40\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000029, v41\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002a, v42\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002b, v43\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002c, v44\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002d, v45\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002e, v46\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00002f, v47\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000030, v48\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000031, v49\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000032, v50\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000033, v51\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000034, v52\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000035, v53\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000036, v54\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000037, v55\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000038, v56\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000039, v57\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003a, v58\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003b, v59\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003c, v60\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003d, v61\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003e, v62\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00003f, v63\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000040, v64\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000041, v65\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000042, v66\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000043, v67\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000044, v68\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000045, v69\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000046, v70\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000047, v71\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000048, v72\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000049, v73\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004a, v74\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004b, v75\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004c, v76\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004d, v77\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004e, v78\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00004f, v79\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000050, v80\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000051, v81\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000052, v82\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000053, v83\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000054, v84\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000055, v85\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000056, v86\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000057, v87\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000058, v88\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a000059, v89\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005a, v90\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005b, v91\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005c, v92\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005d, v93\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005e, v94\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n v_cmp_ne_u32_e32 vcc, 0x5a00005f, v95\n v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n"
               : "+v"(bad) : "s"(iters) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s8", "scc", "vcc");
  unsigned lbad = 0;
  for (int i = threadIdx.x; i < 4096; i += 64) lbad += lds[i] != (0xa5000000u | i);
  if (bad) atomicAdd(&counts[0], bad);
  if (lbad) atomicAdd(&counts[1], lbad);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

// A wave that keeps LOADING: LDS words (b32 and b128 reads) and global words of known content, and counts the values that come back
// wrong - the traffic the library's keypoint kernels have (gathers + LDS windows), without their arithmetic.
__global__ __launch_bounds__(64) void load_kernel(int iters, const uint4 *__restrict__ gpat, int gwords, unsigned *__restrict__ counts) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xa5000000u | i;
  __syncthreads();
  unsigned bad_lds = 0, bad_g = 0;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const unsigned i1 = (h >> 8) & 4095u, i4 = (h >> 12) & 4092u & ~3u, ig = (h >> 5) % (unsigned)gwords;
    const unsigned a = ((volatile unsigned *)lds)[i1];
    asm volatile("" ::: "memory");
    const uint4 b = *(const uint4 *)(lds + i4);
    const uint4 g = gpat[ig];
    // the keypoint kernels' gather: 8 bytes at 4-byte alignment (two horizontally adjacent pixels), any dword position
    struct __attribute__((packed, aligned(4))) Pair { unsigned a, b; };
    const unsigned iu = (h >> 3) % (unsigned)(4 * gwords - 2);
    const Pair pr = *(const Pair *)((const unsigned *)gpat + iu);
    bad_g += (pr.a != iu) + (pr.b != iu + 1u);
    bad_lds += a != (0xa5000000u | i1);
    bad_lds += (b.x != (0xa5000000u | i4)) + (b.y != (0xa5000000u | (i4 + 1))) + (b.z != (0xa5000000u | (i4 + 2))) + (b.w != (0xa5000000u | (i4 + 3)));
    bad_g += (g.x != 4u * ig) + (g.y != 4u * ig + 1u) + (g.z != 4u * ig + 2u) + (g.w != 4u * ig + 3u);
  }
  if (bad_lds) atomicAdd(&counts[1], bad_lds);
  if (bad_g) atomicAdd(&counts[0], bad_g);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

// A wave that keeps COMPUTING in double precision (fma, sqrt, division: what Baumberg's inverse square root, the photometric sums
// and the exported frames use) and compares every round with its first one.
__global__ __launch_bounds__(64) void fp64_kernel(int iters, double seed0, unsigned *__restrict__ counts) {
  double seed = seed0 + 1e-3 * threadIdx.x + 1e-6 * (blockIdx.x & 1023);
  double ref = 0;
  float reff = 0;
  unsigned bad = 0, badf = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(seed));
    double x = seed, acc = 0;
    float xf = (float)seed, accf = 0.f;
#pragma unroll 4
    for (int q = 0; q < 16; q++) {
      x = fma(x, 1.0000001, 0.25);
      const double r = 1.0 / sqrt(1.0 + x * x);
      acc += r * x + (x - acc) / (2.0 + r);
      xf = fmaf(xf, 1.0001f, 0.25f);
      accf += sqrtf(1.0f + xf * xf) / (2.0f + xf);
    }
    if (it == 0) { ref = acc; reff = accf; }
    bad += acc != ref;
    badf += accf != reff;
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (badf) atomicAdd(&counts[1], badf);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

// A wave that keeps running PACKED fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: what the compiler makes of the two
// coordinate chains WX += a11, WY += a21 of the keypoint kernels) and compares every round with its first one.
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void pk_kernel(int iters, float seed0, unsigned *__restrict__ counts) {
  const float sd = seed0 + 1e-3f * threadIdx.x;
  v2f ref = {0.f, 0.f};
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
    v2f x = {sd, sd * 0.5f}, a = {0.37f, 0.21f}, m = {1.0001f, 0.9999f};
    asm volatile("" : "+v"(x), "+v"(a), "+v"(m));
#pragma unroll
    for (int q = 0; q < 32; q++) {
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(a));
    }
    if (it == 0) ref = x;
    bad += (x.x != ref.x) + (x.y != ref.y);
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

// The 2x2 Jacobi SVD of csrc/detect.hip (baumberg_hessian_kernel's fp64 part: products, hypot, sqrt, divisions, conversions) on
// inputs that depend on the lane only, every round compared with the first one.  SVD_PART: 0 whole routine.
__device__ bool svd2x2_f32(const float A[4], float d[2], float U[4], float Vt[4]) {   // false: a singular value <= FLT_MIN
  const float eps = 1.1920929e-7f * 2;
  const double minval = 1.17549435e-38;
  float a0[2] = {A[0], A[2]}, a1[2] = {A[1], A[3]};       // rows of A^T
  float v0[2] = {1.f, 0.f}, v1[2] = {0.f, 1.f};
  double W0 = (double)a0[0] * a0[0] + (double)a0[1] * a0[1];
  double W1 = (double)a1[0] * a1[0] + (double)a1[1] * a1[1];
  for (int iter = 0; iter < 30; iter++) {
    double p = (double)a0[0] * a1[0];
    p += (double)a0[1] * a1[1];
    if (fabs(p) <= eps * sqrt(W0 * W1)) break;
    p *= 2;
    const double beta = W0 - W1, gamma = hypot(p, beta);
    float c, s;
    if (beta < 0) {
      const double delta = (gamma - beta) * 0.5;
      s = (float)sqrt(delta / gamma);
      c = (float)(p / (gamma * s * 2));
    } else {
      c = (float)sqrt((gamma + beta) / (gamma * 2));
      s = (float)(p / (gamma * c * 2));
    }
    W0 = 0; W1 = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const float t0 = c * a0[q] + s * a1[q];
      const float t1 = -s * a0[q] + c * a1[q];
      a0[q] = t0; a1[q] = t1;
      W0 += (double)t0 * t0; W1 += (double)t1 * t1;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const float t0 = c * v0[q] + s * v1[q];
      const float t1 = -s * v0[q] + c * v1[q];
      v0[q] = t0; v1[q] = t1;
    }
  }
  W0 = sqrt((double)a0[0] * a0[0] + (double)a0[1] * a0[1]);
  W1 = sqrt((double)a1[0] * a1[0] + (double)a1[1] * a1[1]);
  if (W0 < W1) {
    const double tw = W0; W0 = W1; W1 = tw;
#pragma unroll
    for (int q = 0; q < 2; q++) { float t = a0[q]; a0[q] = a1[q]; a1[q] = t; t = v0[q]; v0[q] = v1[q]; v1[q] = t; }
  }
  d[0] = (float)W0; d[1] = (float)W1;
  if (W0 <= minval || W1 <= minval) return false;
  const float s0 = (float)(1 / W0), s1 = (float)(1 / W1);
  a0[0] *= s0; a0[1] *= s0; a1[0] *= s1; a1[1] *= s1;
  U[0] = a0[0]; U[1] = a1[0]; U[2] = a0[1]; U[3] = a1[1];
  Vt[0] = v0[0]; Vt[1] = v0[1]; Vt[2] = v1[0]; Vt[3] = v1[1];
  return true;
}

template <int V> __device__ bool svd_var(const float A[4], float d[2], float U[4], float Vt[4]) {   // false: a singular value <= FLT_MIN
  const float eps = 1.1920929e-7f * 2;
  const double minval = 1.17549435e-38;
  float a0[2] = {A[0], A[2]}, a1[2] = {A[1], A[3]};       // rows of A^T
  float v0[2] = {1.f, 0.f}, v1[2] = {0.f, 1.f};
  double W0 = (double)a0[0] * a0[0] + (double)a0[1] * a0[1];
  double W1 = (double)a1[0] * a1[0] + (double)a1[1] * a1[1];
  for (int iter = 0; iter < ((V & 256) ? 1 : (V & 15) == 3 ? 3 : 30); iter++) {
    double p = (double)a0[0] * a1[0];
    p += (double)a0[1] * a1[1];
    if ((V & 15) != 3 && fabs(p) <= eps * sqrt(W0 * W1)) break;
    p *= 2;
    const double beta = W0 - W1, gamma = ((V & 15) == 1 || (V & 16)) ? sqrt(p * p + beta * beta) : hypot(p, beta);
    float c, s;
    if ((V & 64)) { c = 0.8f; s = 0.6f * (float)(p / (gamma + 1.0)); } else if ((V & 15) != 2 && beta < 0) {
      const double delta = (gamma - beta) * 0.5;
      s = (float)sqrt(delta / gamma);
      c = (float)(p / (gamma * s * 2));
    } else {
      c = (float)sqrt((gamma + beta) / (gamma * 2));
      s = (float)(p / (gamma * c * 2));
    }
    W0 = 0; W1 = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const float t0 = c * a0[q] + s * a1[q];
      const float t1 = -s * a0[q] + c * a1[q];
      a0[q] = t0; a1[q] = t1;
      if (V & 128) { W0 = (double)((float)W0 + t0 * t0); W1 = (double)((float)W1 + t1 * t1); } else { W0 += (double)t0 * t0; W1 += (double)t1 * t1; }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (V & 32) continue;
      const float t0 = c * v0[q] + s * v1[q];
      const float t1 = -s * v0[q] + c * v1[q];
      v0[q] = t0; v1[q] = t1;
    }
  }
  if ((V & 15) == 4) { d[0] = (float)W0; d[1] = (float)W1; U[0] = a0[0]; U[1] = a1[0]; U[2] = a0[1]; U[3] = a1[1]; Vt[0] = v0[0]; Vt[1] = v0[1]; Vt[2] = v1[0]; Vt[3] = v1[1]; return true; }
  W0 = sqrt((double)a0[0] * a0[0] + (double)a0[1] * a0[1]);
  W1 = sqrt((double)a1[0] * a1[0] + (double)a1[1] * a1[1]);
  if (W0 < W1) {
    const double tw = W0; W0 = W1; W1 = tw;
#pragma unroll
    for (int q = 0; q < 2; q++) { float t = a0[q]; a0[q] = a1[q]; a1[q] = t; t = v0[q]; v0[q] = v1[q]; v1[q] = t; }
  }
  d[0] = (float)W0; d[1] = (float)W1;
  if (W0 <= minval || W1 <= minval) return false;
  const float s0 = (float)(1 / W0), s1 = (float)(1 / W1);
  a0[0] *= s0; a0[1] *= s0; a1[0] *= s1; a1[1] *= s1;
  U[0] = a0[0]; U[1] = a1[0]; U[2] = a0[1]; U[3] = a1[1];
  Vt[0] = v0[0]; Vt[1] = v0[1]; Vt[2] = v1[0]; Vt[3] = v1[1];
  return true;
}

// the same routine with one piece changed (which piece makes it vulnerable?): 1 sqrt(p^2 + beta^2) instead of hypot, 2 one branch of the
// rotation only, 3 three sweeps without the convergence test, 4 without the final square roots / ordering / normalisation
template <int V>
__global__ __launch_bounds__(256) void svdvar_kernel(int iters, float seed0, unsigned *__restrict__ counts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float A[4] = {seed0 + 0.013f * (t & 1023), 0.3f + 0.001f * (t & 255), 0.3f + 0.001f * (t & 255), -1.7f + 0.007f * (t & 511)};
  float rd[2] = {0, 0}, rU[4] = {0, 0, 0, 0}, rV[4] = {0, 0, 0, 0};
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]));
    float d[2], U[4], Vt[4];
    const bool ok = svd_var<V>(A, d, U, Vt);
    if (!ok) { d[0] = d[1] = 0; for (int q = 0; q < 4; q++) U[q] = Vt[q] = 0; }
    if (it == 0) { rd[0] = d[0]; rd[1] = d[1]; for (int q = 0; q < 4; q++) { rU[q] = U[q]; rV[q] = Vt[q]; } }
    unsigned b = (d[0] != rd[0]) + (d[1] != rd[1]);
    for (int q = 0; q < 4; q++) b += (U[q] != rU[q]) + (Vt[q] != rV[q]);
    bad += b != 0;
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

__global__ __launch_bounds__(256) void svd_kernel(int iters, float seed0, unsigned *__restrict__ counts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float A[4] = {seed0 + 0.013f * (t & 1023), 0.3f + 0.001f * (t & 255), 0.3f + 0.001f * (t & 255), -1.7f + 0.007f * (t & 511)};
  float rd[2] = {0, 0}, rU[4] = {0, 0, 0, 0}, rV[4] = {0, 0, 0, 0};
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]));
    float d[2], U[4], Vt[4];
    const bool ok = svd2x2_f32(A, d, U, Vt);
    if (!ok) { d[0] = d[1] = 0; for (int q = 0; q < 4; q++) U[q] = Vt[q] = 0; }
    if (it == 0) { rd[0] = d[0]; rd[1] = d[1]; for (int q = 0; q < 4; q++) { rU[q] = U[q]; rV[q] = Vt[q]; } }
    unsigned b = (d[0] != rd[0]) + (d[1] != rd[1]);
    for (int q = 0; q < 4; q++) b += (U[q] != rU[q]) + (Vt[q] != rV[q]);
    bad += b != 0;
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

// Parts of that routine on their own: 1 hypot, 2 float <-> double conversions around a product, 3 sqrt and division, 4 fma / add / mul
// only, 5 comparisons and selects on doubles, 6 v_rcp_f64 / v_rsq_f64 alone (the hardware approximations)
template <int PART>
__global__ __launch_bounds__(256) void part_kernel(int iters, float seed0, unsigned *__restrict__ counts) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a = seed0 + 0.013f * (t & 1023), b = -1.7f + 0.007f * (t & 511);
  double ref = 0;
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(a), "+v"(b));
    double r = 0;
#pragma unroll 4
    for (int q = 0; q < 8; q++) {
      const double x = (double)a + 0.125 * q, y = (double)b - 0.25 * q;
      if (PART == 1) r += hypot(x, y);
      else if (PART == 2) { const float f = (float)(x * y); r += (double)f * (double)(a + (float)q); }
      else if (PART == 3) r += sqrt(x * x + 1.0) / (2.0 + y * y);
      else if (PART == 4) r = fma(r, 1.0000001, x * y) + (x - y);
      else if (PART == 5) r += (fabs(x) <= 1.5 * fabs(y)) ? (x < y ? x : y) : (x > -y ? 0.5 : y);
      else if (PART == 6) { r += __builtin_amdgcn_rcp(x * x + 1.0) + __builtin_amdgcn_rsq(y * y + 2.0); }
      else { r += x * y; }
    }
    if (PART >= 14) {      // the rotation block of the SVD on its own: fp64 sqrt / division / conversion to float on both sides of a
                           // data-dependent branch.  14: as in the routine after convergence (p = 0 exactly), 15: p small but not 0,
                           // 16: the same arithmetic without the branch (one side), 17: zero numerators only: 0 / x in fp64
      double acc = 0;
#pragma unroll 2
      for (int q = 0; q < 4; q++) {
        double pp = PART == 14 ? 0.0 : (double)a * 1e-9 * (q + 1);
        double beta = ((t + q) & 1) ? (double)b - 0.5 * q : -(double)b + 0.25 * q;
        asm volatile("" : "+v"(pp), "+v"(beta));
        if (PART == 17) { acc += (pp * 0.0) / (beta * beta + 2.0) + 0.0 / (beta + 3.0); continue; }
        pp *= 2;
        const double gamma = sqrt(pp * pp + beta * beta);
        float c, sn;
        if (PART != 16 && beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          sn = (float)sqrt(delta / gamma);
          c = (float)(pp / (gamma * sn * 2));
        } else {
          c = (float)sqrt((gamma + beta) / (gamma * 2));
          sn = (float)(pp / (gamma * c * 2));
        }
        acc += (double)c * 3.0 + (double)sn;
      }
      r += acc;
    } else if (PART >= 10) {      // DENORMAL operands and results: 10 fp32 products / sums, 11 double -> float conversions into the denormal range,
                           // 12 fp64 products in the denormal range, 13 float rotations (c * x + s * y) with a denormal s, as the SVD's later sweeps
      float tiny = a * 1e-30f, acc = 0.f;
      double dt = (double)a * 1e-300, dacc = 0.0;
      asm volatile("" : "+v"(tiny), "+v"(dt));
#pragma unroll 4
      for (int q = 1; q <= 8; q++) {
        if (PART == 10) { const float y = tiny * (1e-10f * q); acc += y * 3.0f + y; }
        else if (PART == 11) { const float y = (float)((double)a * 1e-40 * q); acc += y; dacc += (double)y * 1e30; }
        else if (PART == 12) { const double y = dt * (1e-12 * q); dacc += y * 3.0 + y; }
        else { const float sn = tiny * 1e-9f * q, cs = 1.0f; const float t0 = cs * a + sn * b, t1 = -sn * a + cs * b; acc += t0 * 1e-3f + t1; dacc += (double)t0 * t0 + (double)t1 * t1; }
      }
      r += (double)acc * 1e38 + dacc * (PART == 12 ? 1e300 : 1.0);
    } else if (PART == 9) {       // plain fp64 work under a lane-dependent trip count and a data-dependent exit: partial EXEC masks
      const int n = 1 + (t & 15);
      double z = (double)a;
      for (int q = 0; q < n; q++) {
        z = fma(z, 1.0000001, (double)b * 0.125) + sqrt(1.0 + z * z) / (2.0 + z * z);
        if (z > 40.0 + (t & 3)) break;
      }
      r += z;
    } else if (PART >= 7) {       // 7: the same plain fp64 work (fma, sqrt, division) on MANY live doubles: registers up to ~v60 carry fp64 operands
      constexpr int N = PART == 7 ? 22 : (PART == 8 ? 10 : 2);
      double v[N];
#pragma unroll
      for (int q = 0; q < N; q++) v[q] = (double)a * (1.0 + 0.01 * q) + (double)b;
#pragma unroll
      for (int rep = 0; rep < 3; rep++) {
#pragma unroll
        for (int q = 0; q < N; q++) v[q] = fma(v[q], 1.0000001, v[(q + 7) % N] * 0.125) + sqrt(1.0 + v[(q + 3) % N] * v[(q + 3) % N]) / (2.0 + v[(q + 5) % N] * v[(q + 5) % N]);
      }
#pragma unroll
      for (int q = 0; q < N; q++) r += v[q];
    }
    if (it == 0) ref = r;
    bad += r != ref;
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (threadIdx.x == 0) atomicAdd(&counts[2], 1u);
}

static hipStream_t g_stream;
static unsigned *g_counts;
extern "C" int load_launch(int blocks, int iters, unsigned *out3);
extern "C" int spin_launch(int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, g_stream, iters, g_counts);
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int fp64_launch(int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  hipLaunchKernelGGL(fp64_kernel, dim3(blocks), dim3(64), 0, g_stream, iters, 0.5, g_counts);
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int pk_launch(int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  hipLaunchKernelGGL(pk_kernel, dim3(blocks), dim3(64), 0, g_stream, iters, 0.5f, g_counts);
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int svd_launch(int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  hipLaunchKernelGGL(svd_kernel, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts);
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int part_launch(int part, int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
  switch (part) {
    case 1: hipLaunchKernelGGL(part_kernel<1>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 2: hipLaunchKernelGGL(part_kernel<2>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3: hipLaunchKernelGGL(part_kernel<3>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 4: hipLaunchKernelGGL(part_kernel<4>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 5: hipLaunchKernelGGL(part_kernel<5>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 6: hipLaunchKernelGGL(part_kernel<6>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 7: hipLaunchKernelGGL(part_kernel<7>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 14: hipLaunchKernelGGL(part_kernel<14>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 15: hipLaunchKernelGGL(part_kernel<15>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 16: hipLaunchKernelGGL(part_kernel<16>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 17: hipLaunchKernelGGL(part_kernel<17>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 10: hipLaunchKernelGGL(part_kernel<10>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 11: hipLaunchKernelGGL(part_kernel<11>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 12: hipLaunchKernelGGL(part_kernel<12>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 13: hipLaunchKernelGGL(part_kernel<13>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 9: hipLaunchKernelGGL(part_kernel<9>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    default: hipLaunchKernelGGL(part_kernel<8>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
  }
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int svdvar_launch(int variant, int blocks, int iters, unsigned *out3) {
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
  switch (variant) {
    case 1: hipLaunchKernelGGL(svdvar_kernel<1>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 2: hipLaunchKernelGGL(svdvar_kernel<2>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3: hipLaunchKernelGGL(svdvar_kernel<3>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 4: hipLaunchKernelGGL(svdvar_kernel<4>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 16: hipLaunchKernelGGL(svdvar_kernel<3 + 16>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 32: hipLaunchKernelGGL(svdvar_kernel<3 + 32>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 64: hipLaunchKernelGGL(svdvar_kernel<3 + 64>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 128: hipLaunchKernelGGL(svdvar_kernel<3 + 128>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 256: hipLaunchKernelGGL(svdvar_kernel<3 + 256>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    case 3 + 16 + 32 + 64: hipLaunchKernelGGL(svdvar_kernel<3 + 16 + 32 + 64>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
    default: hipLaunchKernelGGL(svdvar_kernel<0>, dim3(blocks), dim3(256), 0, g_stream, iters, 0.5f, g_counts); break;
  }
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}

extern "C" int load_launch(int blocks, int iters, unsigned *out3) {
  static uint4 *gpat = nullptr;
  const int gwords = 1 << 20;       // 16 MB of uint4 whose content is its own word index
  if (!g_counts) {
    if (hipMalloc(&g_counts, 16) != hipSuccess) return 1;
    if (hipMemset(g_counts, 0, 16) != hipSuccess) return 1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
  }
  if (!gpat) {
    if (hipMalloc(&gpat, (size_t)gwords * 16) != hipSuccess) return 1;
    unsigned *h = (unsigned *)malloc((size_t)gwords * 16);
    for (size_t i = 0; i < (size_t)gwords * 4; i++) h[i] = (unsigned)i;
    if (hipMemcpy(gpat, h, (size_t)gwords * 16, hipMemcpyHostToDevice) != hipSuccess) return 1;
    free(h);
  }
  hipLaunchKernelGGL(load_kernel, dim3(blocks), dim3(64), 0, g_stream, iters, gpat, gwords, g_counts);
  if (hipGetLastError() != hipSuccess) return 3;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return 4;
  if (hipMemcpy(out3, g_counts, 12, hipMemcpyDeviceToHost) != hipSuccess) return 5;
  return 0;
}
