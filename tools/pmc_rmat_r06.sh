#!/bin/bash
# Round 6: memory-side counters of y = A_hat x at BASELINE config 5's size (RMAT 10M / 200M, d = 128), one counter group per rocprofv3 pass
# (tools/pmc_run.sh), for the product's launch policy and for the alternatives of tools/spmm_rmat_sweep.py -- why the L2 hit rate does (not) move:
#   base  the product: 2 time-separated 256-B slices, H = 65,536 rows declared hot, every other row fetched non-temporally
#   p4    4 slices of 128 B pinned to XCDs (slice = workgroup id mod 4), H = 65,536
#   t4    4 time-separated slices of 128 B, H = 131,072
#   fwd1  the product's policy with the Hadamard epilogue (the bench line's `roofline` kernel)
# usage: pmc_rmat_r06.sh [cases...]   -> gpurun_out/pmc/r06_rmat_<case>_<group> -> tools/pmc_pack_rmat_r06.py -> profiles/r06_spmm_pmc_rmat10m*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
N=10000000; M=200000000
cases=${@:-base p4 t4 fwd1}
for c in $cases; do
  case $c in
    base) opts=""; mode=base ;;
    p4)   opts="spmm_slices=4,spmm_pin=1,spmm_hot_rows=65536"; mode=base ;;
    t4)   opts="spmm_slices=4,spmm_hot_rows=131072"; mode=base ;;
    fwd1) opts=""; mode=fwd1 ;;
    *) echo "unknown case $c"; exit 2 ;;
  esac
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
    t=$(echo $grp | cut -d' ' -f1)
    GSS_OPTIONS="$opts" PMC_TIMEOUT=400 bash $R/tools/pmc_run.sh "r06_rmat_${c}_$t" "$grp" tools/spmm_two_pass.py $N $M 128 $mode 3 | grep -v "^$"
  done
done
