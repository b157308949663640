#!/bin/bash
# how large is the placement bias of tools/ab_inproc.py? identical knob values on both plans, three processes
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4am; mkdir -p $O; cd $R
for k in 1 2 3; do
timeout -k 10 300 python3 tools/ab_inproc.py spmm_pair 1 1 full 8 200 2>&1 | grep -E "difference" 
done
for k in 1 2; do
timeout -k 10 300 python3 tools/ab_inproc.py gemm_lines 1 1 full 8 200 2>&1 | grep -E "difference"
done
