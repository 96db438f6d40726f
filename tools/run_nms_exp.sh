cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
L=gpurun_out/r03b/nms_exp.log; : > $L
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export MODS_LIB=$GRAFT_REPO_ROOT/mods-light-zmq_amd/_variants/libmodsgpu_$v.so; else unset MODS_LIB; fi
  echo "== variant '$v'" >> $L
  python tools/prof_detect.py 16 2>&1 | grep -v "^W\|^E\|amdgpu.ids" >> $L
done
unset MODS_LIB
echo "== no side stream" >> $L
MODS_NO_SIDE_STREAM=1 python tools/prof_detect.py 16 2>&1 | grep -v "^W\|^E\|amdgpu.ids" >> $L
echo "== batch 2" >> $L
python tools/prof_detect.py 2 2>&1 | grep -v "^W\|^E\|amdgpu.ids" >> $L
(timeout 900 python -m pytest tests/test_gpu_detect.py tests/test_gpu_describe.py tests/test_gpu_pair.py tests/test_gpu_views.py -x -q -m gpu 2>&1 | tail -3) >> $L
