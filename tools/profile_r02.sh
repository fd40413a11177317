#!/bin/bash
# Round-2 evidence run (GPU box): rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes over bench.py
# (tools/profile_bench.sh), then six 4-counter SQ passes over the headline prefill launch (tools/pmc_prefill.sh).
# Summaries land under gpurun_out/; copy what is to be judged into profiles/r02_*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
tail -n 12 gpurun_out/profile_bench.log
bash tools/pmc_prefill.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16" > gpurun_out/pmc_prefill.log 2>&1
tail -n 40 gpurun_out/pmc_prefill.log
