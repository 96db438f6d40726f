# round 4: the matcher's evidence files (copied to profiles/r04_* by hand afterwards)
#   r04_match_variants.log        parity tests + kernel times, shipped library against mods-light-zmq_amd/_variants/* (the round-3 matcher)
#   r04_match_pmc.txt             SQ / GRBM counters of match_nn1_kernel on the configs[4]-sized lists
#   r04_mfma_fp64_disturbance.log what disturbs the fp64 victim: pure MFMA streams by pattern, and the library's contexts
#   r04_mfma_i8_ubench.log        what v_mfma_i32_32x32x32_i8 sustains (clock, busy) by occupancy / chains / data
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
bash tools/run_match_r04.sh > $O/r04_match_variants.log 2>&1
bash tools/pmc_match_r04.sh > /dev/null 2>&1; cp gpurun_out/r04_match/pmc_libmodsgpu.txt $O/r04_match_pmc.txt
( echo "# SPIN_SVD victim (tools/ubench/spin_victim.hip) next to ONE stream of tools/ubench/mfma_aggr.hip; mode = dtype*100 + chains*10 + kind"
  echo "#   kind 0 carried accumulators, 1 re-seeded + VALU max, 2 chain-major, 4 s_nop 1 between MFMAs, 7 a VALU op between MFMAs,"
  echo "#   9 = kind 1 in 512-register waves (own their SIMD), 11 = kind 1 in 256-register waves of 4-wave workgroups (do not)"
  MODES="10 11 110 20 21 22 24 27 40 41 140 210 240 49 29 411" DATAS="1 0" bash tools/run_mfma_aggr.sh
  echo "# the library: SVD victim and the victim context of test_contexts_on_one_gpu_do_not_disturb_each_other next to three contexts"
  for a in match pair mser view; do echo "aggressor contexts run: $a"; AGGR=$a NO_PYTEST=$([ $a = match ] || echo 1) bash tools/run_dist_r04.sh; done ) > $O/r04_mfma_fp64_disturbance.log 2>&1
bash tools/ubench/run_mfma_pmc.sh > $O/r04_mfma_i8_ubench.log 2>&1
