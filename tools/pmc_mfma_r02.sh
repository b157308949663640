#!/bin/bash
# Round 2: MFMA counters of every matrix kernel of a training step (config 2), one rocprofv3 --pmc pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
PMC_FILTER=kernel bash $R/tools/pmc_run.sh r02_mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" bench.py --steps 5 --warmup 2 --no-cpu-baseline --min-time 0 --spinup-time 0
