#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ao; mkdir -p $O; cd $R
timeout -k 10 1000 bash tools/knob_sweep_live.sh > $O/knob_sweep_live.txt 2>&1; echo "rc=$?"; cat $O/knob_sweep_live.txt
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
