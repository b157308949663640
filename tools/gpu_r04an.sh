#!/bin/bash
# round 4: the kept kernel changes re-measured on ONE live plan (no placement bias): null checks first
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4an; mkdir -p $O; cd $R
run() { timeout -k 10 300 python3 tools/ab_live.py "$@" 2>&1 | grep -E "difference|mean" | grep -v amdgpu; }
{
echo "# null checks (identical values): the tool's own bias"
run spmm_pair 1 1 full 12 300
run gemm_lines 1 1 full 12 300
run gemm_lines 1 1 full 12 300
echo "# the kept changes, old value minus new value (positive = the change is a gain)"
run gemm_lines 0 1 full 12 300
run gemm_variant 4 2 full 12 300
run spmm_pair 0 1 full 12 300
run wgrad_deep 0 2 full 12 300
run gemm_hoist 0 1 full 12 300
echo "# the trainer's step"
run gemm_lines 0 1 lazy_kept 12 300
run gemm_variant 4 2 lazy_kept 12 300
run spmm_pair 0 1 lazy_kept 12 300
run wgrad_deep 0 2 lazy_kept 12 300
run gemm_hoist 0 1 lazy_kept 12 300
} > $O/ab_live.txt 2>&1
cat $O/ab_live.txt
