#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ak; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -8 $O/pytest.txt; echo "pytest rc=$rc"
