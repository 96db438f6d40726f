cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
(for m in poison1 poison2 poison3; do for pat in 7fc00000 ffffffff 7f7f7f7f; do echo "== aggressors: $m pattern $pat (1 registers, 2 LDS, 3 both)"; timeout 600 python tools/stress_match.py 3 1500 $m $pat 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tail -3; done; done) > gpurun_out/r03b/poison.log 2>&1
