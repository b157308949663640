#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4al; mkdir -p $O; cd $R
timeout -k 10 1000 bash tools/knob_sweep_inproc.sh > $O/knob_sweep.txt 2>&1; echo "rc=$?"; grep difference $O/knob_sweep.txt
