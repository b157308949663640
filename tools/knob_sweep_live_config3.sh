# every live kernel-selection knob against its default at config 3 (whole_graph_pathway, d = 256, L = 3) on ONE live plan; run from the repo root on the GPU box
for spec in "gemm_variant 2 5" "gemm_hoist 1 0" "xcd_remap 1 0" "wgrad_variant 1 2" "wgrad_deep 2 1" "spmm_fly 4 8" "spmm_pair 1 0" "gemm_small_nt 2 4" "gemm_small_nt 2 0"; do
  python3 tools/ab_live.py $spec full 6 120 3 2>&1 | grep -E "difference" | grep -v amdgpu
done
