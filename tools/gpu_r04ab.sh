#!/bin/bash
# round 4, final evidence pass 1: the whole GPU suite, then tools/profile_r04.sh (bench line, rocprofv3 full / lazy, PMC of the two forward SpMMs) and the step by class
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ab; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -5 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash tools/profile_r04.sh > $O/profile_r04.log 2>&1; echo "profile_r04 rc=$?"; tail -4 $O/profile_r04.log
python3 tools/lazy_step_prof.py > $O/lazy_step_classes.txt 2>&1; cat $O/lazy_step_classes.txt
