#!/bin/bash
# Round 4, item 3: memory-side counters of y = A_hat x at BASELINE config 5's size (RMAT 10M / 200M, d = 128) -- the single pass the
# product runs and the two-pass column split of tools/spmm_two_pass.py -- one counter group per rocprofv3 pass.
#   usage: pmc_rmat_r04.sh <nodes> <edges> <H> <slices>      -> gpurun_out/pmc/r04_rmat_{base,two}_*; tools/pmc_pack_rmat_r04.py packs them
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
N=${1:-10000000}; M=${2:-200000000}; H=${3:-65536}; NS=${4:-8}
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo $grp | cut -d' ' -f1)
  PMC_TIMEOUT=500 bash $R/tools/pmc_run.sh "r04_rmat_base_$t" "$grp" tools/spmm_two_pass.py $N $M 128 base 3 | grep -v "^$"
  PMC_TIMEOUT=500 bash $R/tools/pmc_run.sh "r04_rmat_two_$t" "$grp" tools/spmm_two_pass.py $N $M 128 two $H $NS 3 | grep -v "^$"
  PMC_TIMEOUT=500 bash $R/tools/pmc_run.sh "r04_rmat_fwd1_$t" "$grp" tools/spmm_two_pass.py $N $M 128 fwd1 3 | grep -v "^$"
done
python3 $R/tools/pmc_pack_rmat_r04.py $N $M $H $NS
