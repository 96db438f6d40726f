# round 4: parity of the matcher, kernel times, and the disturbance checks, in one call
cd $GRAFT_REPO_ROOT
bash tools/run_match_r04.sh 2>&1 | grep "passed\|failed\|==\|nn1\|^C\|fix\|fginn"
NO_PYTEST= bash tools/run_dist_r04.sh
