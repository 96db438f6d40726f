# round 4: SQ / GRBM counters of the matcher kernels on the configs[4]-sized lists (MODS_LIB may select a variant)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_match
mkdir -p $OUT $R/tools/_cache
[ -f $R/tools/_cache/match_fixture.npz ] || python $R/tools/bench_match.py --make > /dev/null 2>&1
for v in ${LIBS:-libmodsgpu.so}; do
  L=$R/mods-light-zmq_amd/$v
  tag=$(basename $v .so)
  rm -rf /tmp/p1 /tmp/p2
  MODS_LIB=$L timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
  MODS_LIB=$L timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
  ( python3 $R/tools/pmc_summary.py $(find /tmp/p1 -name "*counter_collection.csv" | head -1) ${KFILTER:-match_nn1}; python3 $R/tools/pmc_summary.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) ${KFILTER:-match_nn1} ) > $OUT/pmc_$tag.txt
  echo "== $tag"; cat $OUT/pmc_$tag.txt
done
