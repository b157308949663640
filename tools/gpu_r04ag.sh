#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ag; mkdir -p $O; cd $R
for v in 512 1024 384; do
timeout -k 10 300 python3 tools/ab_inproc.py loss_wgs 256 $v full 8 300 > $O/ab_loss_wgs_$v.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_loss_wgs_$v.txt
done
