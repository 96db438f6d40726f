# GPU box: the non-headline configurations (BASELINE configs[2], [3], [4]) through bench.py; usage: bash tools/run_configs.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
( echo "# python bench.py --config c3|c4|c5 --no-cpu-baseline on one MI355X (BASELINE configs[2], [3], [4])"
  python bench.py --config c3 --ladder hessian --no-cpu-baseline 2>/dev/null | grep '^{"metric"'
  python bench.py --config c3 --ladder full --no-cpu-baseline 2>/dev/null | grep '^{"metric"'
  for c in c4 c5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{"metric"'; done
  python bench.py --config c5 --scene two_planes --no-cpu-baseline 2>/dev/null | grep '^{"metric"' ) > gpurun_out/$TAG/configs_c3_c4_c5.log
cut -c1-420 gpurun_out/$TAG/configs_c3_c4_c5.log
