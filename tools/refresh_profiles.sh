# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh r02'): the bench line, the rocprofv3 kernel statistics of the same
# command, the isolated detector leg (one stream, 32 images per launch = the bench's batching since round 6), the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate
# runs, --kernel-trace only) for the HBM traffic of the fused blur + response kernel at the bench's batching, the matcher
# micro-benchmark (kernel statistics + SQ counters), and SQ counters of the describe-stage kernels.  Outputs: gpurun_out/<tag>/.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
grep '^{"metric"' $OUT/stats.log > $OUT/bench_under_rocprof.json
# the isolated leg alone (one stream, 32 images per launch): its blur durations are the ones bench.py's "isolated" events see
PYR_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 32 > $OUT/detect_leg.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/detect_leg_kernel_stats.csv
# the same leg as shipped (small octaves on the prioritised side stream: launches of the two streams overlap)
rm -rf $OUT/leg; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 32 > $OUT/detect_leg_two_streams.log 2>&1
cp $(find $OUT/leg -name "*kernel_stats.csv" | head -1) $OUT/detect_leg_two_streams_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/tools/prof_detect.py 32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/tools/prof_detect.py 32 > /dev/null 2>&1
python3 $R/tools/pmc_blur.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_blur_traffic.csv
# matcher: BASELINE configs[4]-sized lists (tools/_cache/match_fixture.npz travels with the snapshot when it exists)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mstats -- python $R/tools/bench_match.py > $OUT/match.log 2>&1
cp $(find $OUT/mstats -name "*kernel_stats.csv" | head -1) $OUT/match_kernel_stats.csv
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/mpmc1 -- python $R/tools/bench_match.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mpmc2 -- python $R/tools/bench_match.py > /dev/null 2>&1
( python3 $R/tools/pmc_summary.py $(find $OUT/mpmc1 -name "*counter_collection.csv" | head -1) match_; python3 $R/tools/pmc_summary.py $(find $OUT/mpmc2 -name "*counter_collection.csv" | head -1) match_ ) > $OUT/match_pmc.txt
# describe stage + detector: SQ counters per kernel (occupancy, VALU busy, LDS bank conflicts, waits) and L2 hit rate
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/dpmc1 -- python $R/tools/prof_describe.py > $OUT/describe_leg.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/dpmc2 -- python $R/tools/prof_describe.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/dpmc3 -- python $R/tools/prof_describe.py > /dev/null 2>&1
( for d in dpmc1 dpmc2 dpmc3; do python3 $R/tools/pmc_summary.py $(find $OUT/$d -name "*counter_collection.csv" | head -1); done ) > $OUT/describe_pmc.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -- python $R/tools/prof_describe.py > /dev/null 2>&1
cp $(find $OUT/dstats -name "*kernel_stats.csv" | head -1) $OUT/describe_leg_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/leg $OUT/mstats $OUT/mpmc1 $OUT/mpmc2 $OUT/dpmc1 $OUT/dpmc2 $OUT/dpmc3 $OUT/dstats
cut -c1-300 $OUT/bench.json
head -3 $OUT/pmc_blur_traffic.csv
cat $OUT/match.log | grep "C5\|C2"
