cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
rm -rf gpurun_out/r03b/nmsprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03b/nmsprof -o det -- python tools/prof_detect.py 16 > gpurun_out/r03b/nmsprof.log 2>&1
f=$(find gpurun_out/r03b/nmsprof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r03b/nms_kernel_stats.csv
find gpurun_out/r03b/nmsprof -name "*.db" -delete; find gpurun_out/r03b/nmsprof -name "*trace*" -delete
