#!/bin/bash
# round 4: forward projection's epilogue in whole 128-B lines (knob gemm_lines), in-process A/B + the GPU tests under the knob
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/ab_inproc.py gemm_lines 0 1 full 12 300 > $O/ab_full.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_full.txt
timeout -k 10 300 python3 tools/ab_inproc.py gemm_lines 0 1 lazy 12 300 > $O/ab_lazy.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_lazy.txt
GSS_OPTIONS=gemm_lines=1 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_dist.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
