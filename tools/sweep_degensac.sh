# Runs on the GPU box (gpurun -- 'bash tools/sweep_degensac.sh [threads ...]'): BASELINE configs[4] (4096 x 4096 planar pair) with
# DEGENSAC's host pool at the given sizes, the verifier's own breakdown (MODS_RANSAC_PROF) and the one-pair latency.
OUT=gpurun_out/degensac; mkdir -p $OUT
for t in ${@:-16 24}; do
  echo threads $t
  MODS_RANSAC_THREADS=$t MODS_RANSAC_PROF=1 python bench.py --config c5 --no-cpu-baseline > $OUT/c5_$t.json 2> $OUT/c5_$t.err
  grep ransacF $OUT/c5_$t.err | tail -2
  python3 -c "
import json
d=json.loads(open('$OUT/c5_$t.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], c['one_pair_at_a_time'], c['stage_ms'], c['pipeline_results_equal_serial'], c['inliers'])"
done
