#!/bin/bash
# blur / pyramid kernel statistics of the detect leg (16 images per launch) for library variants: tools/run_blur_variants.sh <out> <variant|X=1>...
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [[ "$v" == *=* ]]; then L=""; else L=$R/mods-light-zmq_amd/_variants/libmodsgpu_$v.so; fi
  rm -rf $OUT/leg
  PYR_STREAMS=${PYR_STREAMS:-2} MODS_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -- python $R/tools/prof_detect.py 16 > $OUT/leg_$v.log 2>&1
  echo "== $v" >> $OUT/variants.log
  grep -a "pyramid \|blur  \|nms " $OUT/leg_$v.log | head -4 >> $OUT/variants.log
  python3 - $(find $OUT/leg -name "*kernel_stats.csv" | head -1) >> $OUT/variants.log <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gauss_blur_fast_kernel" in r["Name"] and ", 2, true" in r["Name"]:
        print("   %-52s calls %4s avg %7.1f us" % (r["Name"].split("(")[0][-52:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf $OUT/leg
cat $OUT/variants.log
