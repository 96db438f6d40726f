# round 4: which builds of the matcher disturb fp64 work of other contexts?  For every library (the shipped one and
# mods-light-zmq_amd/_variants/*): the SVD victim of tools/ubench/spin_victim.hip next to three matcher contexts (wrong rounds),
# and the victim context of tests/test_gpu_pair.py::test_contexts_on_one_gpu_do_not_disturb_each_other
cd $GRAFT_REPO_ROOT
for v in mods-light-zmq_amd/libmodsgpu.so mods-light-zmq_amd/_variants/libmodsgpu_*.so; do
  echo "== $v"
  MODS_LIB=$GRAFT_REPO_ROOT/$v SPIN_SVD=1 AGGR=${AGGR:-match} timeout 300 python tools/stress_spin.py 3 ${SPIN_LAUNCHES:-300} 2>&1 | grep "aggressor contexts"
  [ -n "$NO_PYTEST" ] || MODS_LIB=$GRAFT_REPO_ROOT/$v timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -m gpu -k "disturb" 2>&1 | grep "AssertionError: \[\|passed\|failed" | cut -c1-200
done
