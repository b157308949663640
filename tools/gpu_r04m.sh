#!/bin/bash
# round 4: weight-gradient reduce with deeper loads / Adam state requested first (knob wgrad_deep), in-process A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py wgrad_deep 0 1 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py wgrad_deep 0 1 lazy 12 300 > $O/ab_lazy.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_lazy.txt
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
