# every KERNEL-SELECTION knob against its default on ONE live plan (tools/ab_live.py: no placement bias); run from the repo root on the GPU box
for spec in "gemm_variant 2 3" "gemm_variant 2 5" "gemm_ws -1 1" "spmm_slices 0 2" "spmm_slices 0 4"; do
  python3 tools/ab_live.py $spec full 8 300 2>&1 | grep -E "difference" | grep -v amdgpu
done
