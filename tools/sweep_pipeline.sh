# development aid: bench.py over pipeline shapes (GPU workers, pairs per batch, verify workers); run on the GPU box
#   bash tools/sweep_pipeline.sh ["workers batch verify" ...]
cd ${GRAFT_REPO_ROOT:-.}
[ $# -eq 0 ] && set -- "6 8 8" "4 8 8" "4 16 8" "3 16 8" "4 24 8" "3 24 8" "3 32 8" "2 32 8" "5 16 8" "6 16 8"
for cfg in "$@"; do
  set -- $cfg
  v=$(python bench.py --no-cpu-baseline --no-match-leg --no-harder-leg --gpu-workers $1 --pairs-per-batch $2 --verify-workers $3 --steps 8 --warmup 2 2>/tmp/sweep.err | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_cpu']['process_cpu_ms_per_pair'], d['latency_ms_single_pair']['pipeline_one_in_flight_host_u8']['median'])")
  echo "gpu workers $1 pairs per batch $2 verify workers $3 -> $v"
  [ -z "$v" ] && tail -2 /tmp/sweep.err | cut -c1-300
done
