cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_match
mkdir -p $OUT $R/tools/_cache
[ -f $R/tools/_cache/match_fixture.npz ] || python $R/tools/bench_match.py --make > /dev/null 2>&1
for v in ${LIBS:-libmodsgpu.so}; do
  L=$R/mods-light-zmq_amd/$v
  tag=$(basename $v .so)
  rm -rf /tmp/p1 /tmp/p2 /tmp/p3
  MODS_LIB=$L timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/p1 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
  MODS_LIB=$L timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
  MODS_LIB=$L timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d /tmp/p3 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
  ( for p in p1 p2 p3; do python3 $R/tools/pmc_summary.py $(find /tmp/$p -name "*counter_collection.csv" | head -1) ${KFILTER:-match_nn1}; done ) > $OUT/pmc_l2_$tag.txt 2>&1
  echo "== $tag"; cat $OUT/pmc_l2_$tag.txt
done
