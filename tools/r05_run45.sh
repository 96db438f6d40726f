#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; cd $R
(timeout 900 python -m pytest tests/test_gpu_detect.py tests/test_gpu_views.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -2)
for v in new rank0 new rank0; do
  L=""; [ $v = rank0 ] && L=$R/mods-light-zmq_amd/_variants/libmodsgpu_rank0.so
  echo "== $v"; MODS_LIB=$L timeout 200 python tools/exp_single_pair.py 2>&1 | grep "streams 2"
  MODS_LIB=$L timeout 300 python bench.py --config c3 --ladder hessian --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  c3 hessian steps', d['ms_per_step'], 'ms')"
done
