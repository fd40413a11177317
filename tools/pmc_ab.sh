#!/bin/bash
# counter passes over one prefill kernel generation of the A/B library: usage  IMPL=.. SHAPE=B,H,S CAUSAL=0|1 tools/pmc_ab.sh <tag> [counter sets ...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export SFA_LIB_PATH=$PWD/starflashattention_amd/lib/libStarFlashAttention_ab.so
tag=$1; shift
if [ $# -eq 0 ]; then
  set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE"
fi
bash tools/pmc_prefill.sh "$@" > gpurun_out/pmc_$tag.txt 2>&1
