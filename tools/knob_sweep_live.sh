# every KERNEL-SELECTION knob against its default on ONE live plan (tools/ab_live.py: no placement bias); run from the repo root on the GPU box
for spec in "gemm_lines 1 0" "gemm_variant 2 3" "gemm_variant 2 5" "spmm_pair 1 0" "wgrad_deep 2 0" "wgrad_deep 2 1" "gemm_hoist 1 0" "xcd_remap 1 0" "wgrad_variant 1 2" "gemm_small_nt 2 1" "gemm_small_nt 2 4" "gemm_small_nt 2 0" "spmm_fly 4 8"; do
  python3 tools/ab_live.py $spec full 8 300 2>&1 | grep -E "difference" | grep -v amdgpu
done
