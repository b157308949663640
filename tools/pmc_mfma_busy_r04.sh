#!/bin/bash
# Round 4: matrix-pipe occupancy of the three fp32-MFMA kernels after the round's changes (one rocprofv3 --pmc pass each, --kernel-trace only):
# SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1,024 SIMDs) against GRBM_GUI_ACTIVE and the wave-level wait / issue shares
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
PMC_FILTER=gemm_nt bash $R/tools/pmc_run.sh r04_gemm_a_d128 "$A" tools/gemm_prof.py 29960 128 10
PMC_FILTER=kernel bash $R/tools/pmc_run.sh r04_wgl_a "$A" tools/wgrad_loss_prof.py 29960 128 10
