#!/bin/bash
# Round 2, after hub-first relabelling (build_shard(relabel="auto") relabels graphs with >= 250k nodes): the same counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "wg_fwd1 whole_graph fwd1 5" "wg_plain whole_graph plain 5" "r1m_relabel rmat:1000000:20000000 plain 3" "r10m_relabel rmat:10000000:200000000 plain 2"; do
  set -- $spec
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag="r02_$1_$(echo $grp | cut -d' ' -f1)"
    bash $R/tools/pmc_run.sh "$tag" "$grp" tools/spmm_prof.py 2 128 $4 $2 $3 | grep -v "^$"
  done
done
