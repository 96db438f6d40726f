cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
(timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -m gpu -k "disturb" 2>&1 | tail -5
echo "== the same test against the build of before the fix (all 128 VGPRs in use)"
MODS_LIB=$GRAFT_REPO_ROOT/mods-light-zmq_amd/_variants/libmodsgpu_base128.so timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -m gpu -k "disturb" 2>&1 | tail -5
python tools/bench_match.py 2>&1 | grep "C5\|C2") > gpurun_out/r03b/hb.log 2>&1
