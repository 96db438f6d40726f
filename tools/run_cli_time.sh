cd /tmp && export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03b/t2; cd $GRAFT_REPO_ROOT/gpurun_out/r03b/t2
R=$GRAFT_REPO_ROOT
for wk in default 1 4; do
  for it in iters_mods.ini iters_ladder.ini; do
    if [ $wk = default ]; then unset MODS_LADDER_WORKERS; else export MODS_LADDER_WORKERS=$wk; fi
    MODS_RANSAC_SEED=4242 $R/mods-light-zmq_amd/mods $R/tests/golden/graf1.png $R/tests/golden/graf6.png o1 o2 k1 k2 m log 0 0 H $R/tests/configs/classic.ini $R/tests/configs/$it > out_${wk}_$it.log 2>&1
    echo "workers $wk $it: $(grep -A1 'Main matching' out_${wk}_$it.log | tail -1)"
  done
done > ../cli_time.log 2>&1
unset MODS_LADDER_WORKERS
cd $R
bash tools/run_ladder_bench.sh
