cd $GRAFT_REPO_ROOT
for cfg in "6 8 8" "8 8 8" "4 8 8" "6 8 12" "6 16 8" "8 16 8" "5 12 10"; do set -- $cfg
python bench.py --no-cpu-baseline --no-match-leg --gpu-workers $1 --pairs-per-batch $2 --verify-workers $3 --steps 8 --pairs-per-step 96 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['config']['stage_ms_per_pair'])"
done
