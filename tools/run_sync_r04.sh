#!/bin/bash
# waits of the pipeline's threads: sleeping polls (default) against the runtime's spinning wait, on all cores and on two
mkdir -p gpurun_out/r04s; O=gpurun_out/r04s
B="python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-match-leg"
for mode in sleep spin sleep:50; do
  echo "== MODS_SYNC=$mode all cores" >> $O/sync.log
  MODS_SYNC=$mode timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host_cpu'], d['config']['stage_ms_per_pair'])" >> $O/sync.log 2>&1
done
for cores in 0-1 0-3; do
  for mode in sleep spin; do
    echo "== MODS_SYNC=$mode taskset $cores" >> $O/sync.log
    MODS_SYNC=$mode timeout 600 taskset -c $cores $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host_cpu'], d['config']['stage_ms_per_pair'])" >> $O/sync.log 2>&1
  done
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1
tail -3 $O/gputest.log
cat $O/sync.log
