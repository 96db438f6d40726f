#!/bin/bash
# describe-leg kernel statistics for variants of the extraction kernels: usage tools/run_extract_variants.sh <out> <lib|env=..>...
ulimit -c 0
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for v in "$@"; do
  echo "== $v" >> $OUT/variants.log
  if [[ "$v" == *=* ]]; then export $v; L=""; else L=$PWD/mods-light-zmq_amd/_variants/libmodsgpu_$v.so; fi
  MODS_LIB=$L SKIP_TESTS=1 bash tools/quick_desc.sh $(basename $OUT)/tmp 2>&1 | grep -i "total\|extract\|big_" >> $OUT/variants.log
  if [[ "$v" == *=* ]]; then unset ${v%%=*}; fi
done
cat $OUT/variants.log
