#!/bin/bash
# round 4: the evidence behind the bench line (tools/profile_r04.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
bash tools/profile_r04.sh > $O/profile_r04.log 2>&1; echo "profile_r04 rc=$?"; tail -12 $O/profile_r04.log
cp profiles/r04_spmm_pmc.json $O/ 2>/dev/null
python3 tools/lazy_step_prof.py > $O/lazy_step_classes.txt 2>&1; cat $O/lazy_step_classes.txt
