# every live kernel-selection knob against its default at config 3 (whole_graph_pathway, d = 256, L = 3) on ONE live plan; run from the repo root on the GPU box
for spec in "gemm_variant 2 5" "gemm_variant 2 3" "spmm_slices 0 2" "spmm_slices 0 8"; do
  python3 tools/ab_live.py $spec full 6 120 3 2>&1 | grep -E "difference" | grep -v amdgpu
done
