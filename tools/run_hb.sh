cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
B=$GRAFT_REPO_ROOT/mods-light-zmq_amd/_variants/libmodsgpu_base128.so
(echo "== loading victims (LDS b32 / b128 reads, global 16-byte gathers of known words) next to 3 contexts running the PRE-FIX match_nn1_kernel only"
SPIN_LOADS=1 MODS_LIB=$B MODS_MATCH_MASK=2 timeout 120 python tools/stress_spin.py 3 600 2>&1 | grep -v amdgpu.ids | tail -2
echo "== loading victims next to 3 contexts running the shipped matcher"
SPIN_LOADS=1 timeout 120 python tools/stress_spin.py 3 600 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/r03b/spin_loads.log 2>&1
