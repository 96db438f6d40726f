#!/bin/bash
# Texture-addresser / vector-L1 counters of the describe-leg kernels (are the gathers of the per-keypoint kernels what bounds them?);
# usage on the GPU box: bash tools/pmc_ta.sh <out tag> [variant]
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
[ -n "$2" ] && export MODS_LIB=$R/mods-light-zmq_amd/_variants/libmodsgpu_$2.so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $R/tools/prof_describe.py > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $f; else echo "pass $i ($set): no counters"; tail -3 $OUT/p$i.log; fi
  rm -rf $OUT/p$i
done > $OUT/ta_pmc.txt
grep -c . $OUT/ta_pmc.txt
