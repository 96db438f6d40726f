cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
(time timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r03b/gpu_suite.log 2>&1
python bench.py > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err
