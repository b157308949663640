#!/bin/bash
# round 4: balanced SpMM with two (col, val) pairs per lane and trip in narrow lane groups (whole-line index loads, knob spmm_pair)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ah; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py spmm_pair 0 1 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py spmm_pair 0 1 lazy_kept 12 300 > $O/ab_kept.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_kept.txt
GSS_OPTIONS=spmm_pair=1 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
