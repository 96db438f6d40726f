# development aid: matcher experiment variants (libmodsgpu_v*.so built with -DMATCH_EXP=...) + PMC passes of the current kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mexp
mkdir -p $OUT
echo "== current"; python $R/tools/bench_match.py 2>&1 | grep "^C[25]"
for v in $R/mods-light-zmq_amd/libmodsgpu_v*.so; do echo "== $v"; MODS_LIB=$v python $R/tools/bench_match.py --c5only 2>&1 | grep "^C5"; done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/mpmc1 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mpmc2 -- python $R/tools/bench_match.py --c5only > /dev/null 2>&1
( python3 $R/tools/pmc_summary.py $(find $OUT/mpmc1 -name "*counter_collection.csv" | head -1) match_nn1; python3 $R/tools/pmc_summary.py $(find $OUT/mpmc2 -name "*counter_collection.csv" | head -1) match_nn1 ) | tee $OUT/match_pmc.txt
rm -rf $OUT/mpmc1 $OUT/mpmc2
