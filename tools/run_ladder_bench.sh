cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
(timeout 900 python -m pytest tests/test_gpu_views.py tests/test_gpu_mser.py tests/test_gpu_cli.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15) > gpurun_out/r03b/ladder_tests.log 2>&1
: > gpurun_out/r03b/c3.log
for wk in 1 2 4; do MODS_LADDER_WORKERS=$wk python bench.py --config c3 --ladder hessian --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/r03b/c3.log; done
MODS_LADDER_WORKERS=4 python bench.py --config c3 --ladder hessian --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/r03b/c3.log
MODS_LADDER_WORKERS=4 python bench.py --config c3 --ladder full --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 >> gpurun_out/r03b/c3.log
