#!/bin/bash
# PROVENANCE of profiles/r06_ab_live_prefetch_and_norm8.txt -- the knobs below (spmm_pre, gemm_variant 6) existed only in the one-off builds that were
# measured (git history: commits of round 6 before "Config 3: the fused-norm last projection as eight waves"); the kept variant is the product's
# default now, the rejected ones are gone.
# Round 6 A/Bs on ONE live plan (tools/ab_live.py): the SpMM's epilogue operands requested ahead of the gathers (spmm_pre 1 = on) and the fused-norm
# d = 256 projection as eight waves (gemm_variant 6); run from the repo root on the GPU box
for spec in "spmm_pre 0 0 full 8 300 2" "spmm_pre 1 0 full 16 300 2" "spmm_pre 1 0 lazy_kept 12 300 2" "spmm_pre 1 0 full 10 120 3" "gemm_variant 2 6 full 10 120 3" "gemm_variant 2 6 lazy 8 120 3"; do
  echo "== ab_live.py $spec"
  python3 tools/ab_live.py $spec 2>&1 | grep -v amdgpu
done
