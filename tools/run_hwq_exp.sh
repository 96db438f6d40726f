cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
L=gpurun_out/r03b/hwq.log; : > $L
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q bench" >> $L
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-match-leg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stage_ms_per_pair'])" >> $L
  for wk in 4 6 8; do
    echo "== GPU_MAX_HW_QUEUES=$q ladder workers $wk" >> $L
    GPU_MAX_HW_QUEUES=$q MODS_LADDER_WORKERS=$wk python bench.py --config c3 --ladder hessian --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $L
  done
done
echo "== 8 queues, 8 gpu workers" >> $L
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-match-leg --gpu-workers 8 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['stage_ms_per_pair'])" >> $L
