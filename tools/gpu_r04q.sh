#!/bin/bash
# round 4: four XCD-pinned 128-B feature slices instead of two of 256 B in the config-2 SpMMs (in-process A/B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
GSS_AB_FIXED=spmm_pin=1 timeout -k 10 300 python3 tools/ab_inproc.py spmm_slices 2 4 full 8 200 > $O/ab_slices4.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_slices4.txt
GSS_AB_FIXED=spmm_pin=1 timeout -k 10 300 python3 tools/ab_inproc.py spmm_slices 0 2 full 8 200 > $O/ab_slices2.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_slices2.txt
