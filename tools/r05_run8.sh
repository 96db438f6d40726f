#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_run8; mkdir -p $OUT
cd $R
timeout 900 python tools/exp_graph.py DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 > $OUT/graph_nopacket.log 2>&1
cat $OUT/graph_nopacket.log
