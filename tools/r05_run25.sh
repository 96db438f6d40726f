#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run25; mkdir -p $OUT
cd $R
MODS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --config c3 --gpus 2 --steps 5 --warmup 1 2> $OUT/c3_2ranks.err | grep '^{"metric"' > $OUT/c3_2ranks.json
cut -c1-1300 $OUT/c3_2ranks.json; grep -i "error" $OUT/c3_2ranks.err | head -3
(timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -3)
bash tools/r05_run24.sh
