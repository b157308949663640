#!/bin/bash
# Round 2: MFMA counters of every matrix kernel of a training step, one rocprofv3 --pmc pass per workload:
#   bash tools/pmc_mfma_r02.sh                                -> gpurun_out/pmc/r02_mfma          (config 2)
#   bash tools/pmc_mfma_r02.sh whole_graph_pathway r02_mfma_config3                              (config 3: d = 256, L = 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-whole_graph}
T=${2:-r02_mfma}
PMC_FILTER=kernel bash $R/tools/pmc_run.sh $T "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --min-time 0 --spinup-time 0
