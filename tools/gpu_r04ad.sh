#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ad; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/gemm_w8_ab.py 29960 250000 1000000 4000000 10000000 > $O/gemm_w8.txt 2>&1; echo "rc=$?"; cat $O/gemm_w8.txt
