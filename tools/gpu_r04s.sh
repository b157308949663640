#!/bin/bash
# round 4: per-wave phases of the projection in the LARGE-N regime (tiles >> slots), 64-node tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/gemm_stamps.py 1000000 128 gemm_variant=4 > $O/stamps_1m.txt 2>&1; echo "rc=$?"; cat $O/stamps_1m.txt
