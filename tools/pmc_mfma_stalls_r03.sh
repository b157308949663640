#!/bin/bash
# Round 3: where the waves of the fp32-MFMA kernels spend their time.  Two rocprofv3 --pmc passes (SQ block: 8 slots) over
# tools/gemm_prof.py (the projection alone, 10 launches) and tools/wgrad_loss_prof.py (weight gradient + loss sweep):
#   pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
#   pass B: SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES SQ_INSTS_MFMA
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVES SQ_INSTS_MFMA"
for d in 128 256; do
  PMC_FILTER=gemm_nt bash $R/tools/pmc_run.sh r03_gemm_a_d$d "$A" tools/gemm_prof.py 29960 $d 10
  PMC_FILTER=gemm_nt bash $R/tools/pmc_run.sh r03_gemm_b_d$d "$B" tools/gemm_prof.py 29960 $d 10
done
PMC_FILTER=kernel bash $R/tools/pmc_run.sh r03_wgl_a "$A" tools/wgrad_loss_prof.py 29960 128 10
PMC_FILTER=kernel bash $R/tools/pmc_run.sh r03_wgl_b "$B" tools/wgrad_loss_prof.py 29960 128 10
