#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ae; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py gemm_variant 2 5 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py gemm_variant 2 5 lazy 12 300 > $O/ab_lazy.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_lazy.txt
timeout -k 10 300 python3 tools/ab_inproc.py gemm_variant 2 5 lazy_kept 12 300 > $O/ab_kept.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_kept.txt
