cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
( echo "# python bench.py --config c3|c4|c5 --no-cpu-baseline on one MI355X (BASELINE configs[2], [3], [4]), round 2"
  for c in c3 c4 c5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{"metric"'; done ) > gpurun_out/r02b/configs_c3_c4_c5.log
cut -c1-260 gpurun_out/r02b/configs_c3_c4_c5.log
python bench.py 2>/dev/null | grep '^{"metric"' > gpurun_out/r02b/bench2.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02b/bench2.json').read())
print(d['value'], d['cpu_baseline'])
PY
