cd $GRAFT_REPO_ROOT
for data in ${DATAS:-1}; do for m in ${MODES:-20 22 23 24 25 26 27 28 40 42 43 44 45 46 47 48}; do
  echo -n "mode $m data $data: "
  AGGR_MFMA=$m AGGR_DATA=$data SPIN_SVD=1 timeout 120 python tools/stress_spin.py ${NAGGR:-1} 200 2>&1 | grep "aggressor contexts"
done; done
