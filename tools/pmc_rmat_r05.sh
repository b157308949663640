#!/bin/bash
# Round 5: memory-side counters of y = A_hat x and of the Hadamard-fused forward product at BASELINE config 5's size (RMAT 10M / 200M,
# d = 128) with this round's kernels (the giant rows chunked: three launches per product, summed) -- one counter group per rocprofv3 pass
#   -> gpurun_out/pmc/r05_rmat_{base,fwd1}_* -> tools/pmc_pack_rmat_r04.py (PMC_ROUND=r05) -> profiles/r05_spmm_pmc_rmat10m.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
N=${1:-10000000}; M=${2:-200000000}
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo $grp | cut -d' ' -f1)
  PMC_TIMEOUT=500 bash $R/tools/pmc_run.sh "r05_rmat_base_$t" "$grp" tools/spmm_two_pass.py $N $M 128 base 3 | grep -v "^$"
  PMC_TIMEOUT=500 bash $R/tools/pmc_run.sh "r05_rmat_fwd1_$t" "$grp" tools/spmm_two_pass.py $N $M 128 fwd1 3 | grep -v "^$"
done
PMC_ROUND=r05 python3 $R/tools/pmc_pack_rmat_r04.py $N $M 65536 8 && cp $R/profiles/r05_spmm_pmc_rmat10m.json $R/gpurun_out/
