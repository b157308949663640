# every plan-level tuning knob against its default, both plans in ONE process (tools/ab_inproc.py); run from the repo root on the GPU box
for spec in "loss_wgs 256 192" "loss_wgs 256 320" "loss_wgs 256 512" "wgrad_wgs 256 192" "wgrad_wgs 256 320" "wgrad_wgs 256 384" "gemm_variant 2 3" "spmm_slices 0 2" "spmm_slices 0 4" "prep_side 1 0" "loss_dgrad 0 1"; do
  python3 tools/ab_inproc.py $spec full 8 200 2>&1 | grep -E "difference|mean" | grep -v amdgpu
done
