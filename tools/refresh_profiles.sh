# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the bench line, the rocprofv3 kernel statistics of the same
# command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) for the HBM traffic of the
# pyramid blur at the bench's batching (16 images per launch).  Outputs under gpurun_out/r01/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > $OUT/bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
grep '^{"metric"' $OUT/stats.log > $OUT/bench_under_rocprof.json
# the roofline leg alone (one stream, 16 images per launch): its blur durations are the ones bench.py's HIP events see
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 16 > $OUT/leg.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/blur_leg_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/tools/prof_detect.py 16 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/tools/prof_detect.py 16 > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_blur.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_blur_traffic.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/leg
cat $OUT/bench.json | cut -c1-300
head -3 $OUT/pmc_blur_traffic.csv
