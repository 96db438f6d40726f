# Runs on the GPU box: the cross-context disturbance experiments of round 3 (DESIGN.md "The matcher and its neighbours").
# Needs mods-light-zmq_amd/_variants/libmodsgpu_base128.so = tools/variant.sh base128 match "-DMATCH_NN1_VGPRS=64" (the kernels of
# before the fix: all 128 VGPRs in use) and tools/ubench/vgpr_top.bin (hipcc --offload-arch=gfx950 -O2 tools/ubench/vgpr_top.hip).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
B=$GRAFT_REPO_ROOT/mods-light-zmq_amd/_variants/libmodsgpu_base128.so
(echo "# tools/run_stress.sh: one victim context repeats detect + describe (+ match) of three 1080p pairs and compares every result with its first"
echo "== BEFORE the fix (match_nn1_kernel / match_fginn_kernel use all 128 VGPRs), 6 contexts run the whole chain"
MODS_LIB=$B timeout 900 python tools/stress_match.py 6 2500 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tail -6
echo "== BEFORE the fix, 1 context alone"
MODS_LIB=$B timeout 900 python tools/stress_match.py 1 6000 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tail -2
for a in detect full match; do echo "== BEFORE the fix, victim + 3 aggressor contexts running: $a"; MODS_LIB=$B timeout 900 python tools/stress_match.py 4 3000 $a 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -3; done
for m in 1 2 4 8 16; do echo "== BEFORE the fix, aggressors run only the matcher kernels of mask $m (1 init+pack, 2 match_nn1_kernel, 4 mid+gather, 8 match_fginn_kernel, 16 emit)"; MODS_LIB=$B MODS_MATCH_MASK=$m timeout 900 python tools/stress_match.py 4 3000 match 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2; done
for a in gemm stream; do echo "== aggressors: torch $a kernels on their own streams"; MODS_LIB=$B MODS_MATCH_MASK=0 timeout 900 python tools/stress_match.py 4 3000 $a 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2; done
echo "== AFTER the fix (<= 124 VGPRs), victim + 3 matcher aggressors"
timeout 900 python tools/stress_match.py 4 6000 match 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2
echo "== AFTER the fix, 6 contexts run the whole chain"
timeout 900 python tools/stress_match.py 6 2500 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2
echo "== tools/ubench/vgpr_top.hip: plain writes to v[124:127] of a 128-VGPR kernel against constants parked in v0..v7 of other waves"
timeout 300 tools/ubench/vgpr_top.bin) > gpurun_out/r03/concurrency_stress.log 2>&1
