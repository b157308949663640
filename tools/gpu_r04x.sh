#!/bin/bash
# round 4: loss sweep with whole-line fetches of the first product's fragments (knob loss_lines), in-process A/B + the loss tests under the knob
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py loss_lines 0 1 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_full.txt
GSS_OPTIONS=loss_lines=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
