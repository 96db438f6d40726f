cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
(time timeout 1400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6) > gpurun_out/r03b/gpu_suite.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03b/smoke.log 2>&1
python bench.py > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err
tail -4 gpurun_out/r03b/gpu_suite.log; tail -1 gpurun_out/r03b/smoke.log; cut -c1-120 gpurun_out/r03b/bench.json
