# development aid: bench.py over pipeline shapes (GPU workers, pairs per batch, pairs per step); run on the GPU box
cd $GRAFT_REPO_ROOT
for cfg in "6 8 48" "6 16 96" "3 16 48" "4 12 48" "8 6 48" "6 8 96" "8 12 96" "4 16 64" "5 8 40" "6 12 72"; do
  set -- $cfg
  v=$(python bench.py --no-cpu-baseline --no-match-leg --gpu-workers $1 --pairs-per-batch $2 --pairs-per-step $3 --steps 8 --warmup 2 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "workers $1 ppb $2 pairs/step $3 -> $v"
done
