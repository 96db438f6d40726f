#!/bin/bash
# describe leg under rocprofv3 for a list of variant libraries (same box): usage r05_run27.sh <variant> ...   ("new" = the built library)
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run27; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  L=""; [ $v != new ] && L=$R/mods-light-zmq_amd/_variants/libmodsgpu_$v.so
  rm -rf /tmp/p_$v
  MODS_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o d -- python $R/tools/prof_describe.py > $OUT/prof_$v.log 2>&1
  f=$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -14 $f > $OUT/describe_leg_$v.csv
  echo "== $v"; grep -E "orient_kernel|extract_small|baumberg|sift_wave2" $OUT/describe_leg_$v.csv | awk -F'",' '{print substr($1,1,40), $2}' | cut -c1-90
done
