#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4aq; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/placement_probe.py 8 6 300 > $O/placement.txt 2>&1; echo "rc=$?"; grep -v amdgpu $O/placement.txt
timeout -k 10 300 python3 tools/placement_probe.py 8 6 300 > $O/placement2.txt 2>&1; echo "rc=$?"; grep -v amdgpu $O/placement2.txt
