#!/bin/bash
# SQ / LDS / clock counters of the conv weight-gradient kernel (kbench shapes), two --pmc passes:  bash tools/pmc_wgrad.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
Bc="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
bash $R/tools/pmc_kernel.sh ${1}a "$A" -- python3 $R/tools/kbench.py wgrad --iters 4 > $R/gpurun_out/r3_pmc_${1}a.txt 2>&1
bash $R/tools/pmc_kernel.sh ${1}b "$Bc" -- python3 $R/tools/kbench.py wgrad --iters 4 > $R/gpurun_out/r3_pmc_${1}b.txt 2>&1
grep -A9 "mfma_wgrad" $R/gpurun_out/r3_pmc_${1}a.txt $R/gpurun_out/r3_pmc_${1}b.txt
rm -rf $R/gpurun_out/pmc_${1}a $R/gpurun_out/pmc_${1}b
