#!/bin/bash
# Round 2: memory-side counters of the SpMM kernels in both regimes, one counter group per rocprofv3 pass
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Run on the GPU box from the repo root:
#   bash tools/pmc_spmm_r02.sh           -> gpurun_out/pmc/<tag>/...
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # tag counters workload mode reps
  bash $R/tools/pmc_run.sh "$1" "$2" tools/spmm_prof.py 2 128 $5 $3 $4 | grep -v "^$"
}
export GSS_RELABEL=0   # the generator / loader node order (round-1 behaviour)
for spec in "wg_norelabel_fwd1 whole_graph fwd1 5" "wg_norelabel_plain whole_graph plain 5" "r1m_plain rmat:1000000:20000000 plain 3" "r10m_plain rmat:10000000:200000000 plain 2"; do
  set -- $spec
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag="r02_$1_$(echo $grp | cut -d' ' -f1)"
    run "$tag" "$grp" $2 $3 $4
  done
done
