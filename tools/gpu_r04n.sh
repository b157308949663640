#!/bin/bash
# round 4: wgrad_deep 1 vs 2 in process, then the evidence refresh (tools/profile_r04.sh) and the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py wgrad_deep 1 2 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py wgrad_deep 1 2 lazy 12 300 > $O/ab_lazy.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_lazy.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -5 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash tools/profile_r04.sh > $O/profile_r04.log 2>&1; echo "profile_r04 rc=$?"; tail -12 $O/profile_r04.log
python3 tools/lazy_step_prof.py > $O/lazy_step_classes.txt 2>&1; cat $O/lazy_step_classes.txt
