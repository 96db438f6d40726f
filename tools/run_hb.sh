cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_detect.py -x -q -m gpu -k "hessian_baumberg or hessian_form or one_view" 2>&1 | tail -25 > gpurun_out/r03b/hb.log
