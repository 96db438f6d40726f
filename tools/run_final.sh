# GPU box, end of a round: the whole GPU suite first, then the profile set, the other configurations and the low-inlier bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; TAG=${1:-r03}; mkdir -p gpurun_out/$TAG
(time timeout 1400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8) > gpurun_out/$TAG/gpu_suite.log 2>&1
bash tools/refresh_profiles.sh $TAG > gpurun_out/$TAG/refresh.out 2>&1
bash tools/run_configs.sh $TAG > /dev/null 2>&1
python bench.py --inlier-ratio 0.4 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/$TAG/bench_inlier_ratio_0.4.json
tail -3 gpurun_out/$TAG/gpu_suite.log; cut -c1-200 gpurun_out/$TAG/bench.json
