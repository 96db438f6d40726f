#!/bin/bash
# SQ counters of the describe-leg kernels (two passes) -> digest; usage: tools/pmc_describe.sh <out tag> [env=value ...]
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
for kv in "$@"; do export $kv; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/dpmc1 -- python $R/tools/prof_describe.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/dpmc2 -- python $R/tools/prof_describe.py > /dev/null 2>&1
( for d in dpmc1 dpmc2; do python3 $R/tools/pmc_summary.py $(find $OUT/$d -name "*counter_collection.csv" | head -1); done ) > $OUT/describe_pmc.txt
rm -rf $OUT/dpmc1 $OUT/dpmc2
python3 $R/tools/pmc_digest.py $OUT/describe_pmc.txt > $OUT/describe_pmc_digest.txt
head -12 $OUT/describe_pmc_digest.txt
