#!/bin/bash
# round 4: feature-split 8-wave projection (knob gemm_fs), in-process A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py gemm_fs 0 1 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py gemm_fs 0 1 lazy 12 300 > $O/ab_lazy.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_lazy.txt
