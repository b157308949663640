#!/bin/bash
# Round 5: matrix-pipe occupancy of the d = 128 forward projection (one rocprofv3 --pmc pass each, --kernel-trace only): the staged tiles at
# config 2's size (N = 29,960) and the weight-stationary persistent kernel where it is the default (N = 250,000 and 1,000,000):
# SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1,024 SIMDs) against GRBM_GUI_ACTIVE and the wave-level wait / issue shares
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
PMC_FILTER=gemm_nt bash $R/tools/pmc_run.sh r05_gemm_d128_n29960 "$A" tools/gemm_prof.py 29960 128 10
PMC_FILTER=proj_ws bash $R/tools/pmc_run.sh r05_projws_d128_n250000 "$A" tools/gemm_prof.py 250000 128 10
PMC_FILTER=proj_ws bash $R/tools/pmc_run.sh r05_projws_d128_n1000000 "$A" tools/gemm_prof.py 1000000 128 6
