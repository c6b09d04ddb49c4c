#!/bin/bash
# GPU box: issue counters of the compiled (variant 11) and the assembly (1035) attention kernel, one counter group per pass
#   bash tools/za_pmc.sh [out_tag]   ->  gpurun_out/pmc_<tag>_{a,b,c}/ + a summary on stdout
TAG=${1:-za}
export ATTN_PLANES=1 ATTN_VARIANTS=${ATTN_VARIANTS:-11,1035}
bash tools/pmc_ops.sh ${TAG}_a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" attn || exit 1
bash tools/pmc_ops.sh ${TAG}_b "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" attn || exit 1
bash tools/pmc_ops.sh ${TAG}_c "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" attn || exit 1
bash tools/pmc_ops.sh ${TAG}_d "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" attn || exit 1
