#!/bin/bash
# round 4: round 3's "nothing moves" knobs (wave priorities, staggered second generation, LDS-footprint occupancy) on one live plan
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ar; mkdir -p $O; cd $R
run() { timeout -k 10 300 python3 tools/ab_live.py "$@" 2>&1 | grep -E "difference" | grep -v amdgpu; }
{
run gemm_prio 0 256 full 8 300
run gemm_prio 0 -1 full 8 300
run gemm_stagger 0 2 full 8 300
run wgrad_prio 0 1 full 8 300
run wgrad_prio 0 2 full 8 300
run loss_lds_kb 0 81 full 8 300
run wgrad_lds_kb 0 81 full 8 300
run gemm_rows_split 1 0 lazy_kept 8 300
} > $O/ab_live3.txt 2>&1
cat $O/ab_live3.txt
