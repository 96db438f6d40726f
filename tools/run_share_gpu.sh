#!/bin/bash
# N ranks of bench.py on ONE GPU over gloo (MODS_BENCH_SHARE_GPU=1), the launcher confined to C cores: what a rank of an
# 8-GPU run on a 16-core host gets from the host side.  usage: tools/run_share_gpu.sh <out tag> <ranks> <cores e.g. 0-7> [bench args]
ulimit -c 0
OUT=gpurun_out/$1; N=$2; CORES=$3; shift 3; mkdir -p $OUT
MODS_BENCH_SHARE_GPU=1 timeout 600 taskset -c $CORES python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 10 --warmup 2 --no-cpu-baseline --no-match-leg "$@" 2>$OUT/err.log | grep '^{"metric"' > $OUT/share_${N}ranks.json
python - $OUT/share_${N}ranks.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["pairs_per_s_by_rank"], d["host"], d["config"]["overlap"], d["host_cpu"]["process_cpu_ms_per_pair"])
PY
