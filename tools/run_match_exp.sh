cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b tools/_cache
python tools/bench_match.py --make > /dev/null 2>&1
bash tools/exp_match.sh > gpurun_out/r03b/match_exp.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_match.py -x -q -m gpu 2>&1 | tail -3) >> gpurun_out/r03b/match_exp.log
