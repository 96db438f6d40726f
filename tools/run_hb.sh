cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 900 python tools/stress_pipeline.py 120 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -12 > gpurun_out/r03b/stress_pipeline.log
