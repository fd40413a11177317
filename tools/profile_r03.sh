#!/bin/bash
# Round-3 evidence run (GPU box).  (1) rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes over the HEADLINE part
# of bench.py only (--no-decode: no supplementary kernels in the trace), statistics over the timed launches only
# (tools/profile_summary.py <dir> <warmup>); (2) six 4-counter SQ passes over the headline prefill launch and over the
# configs[4] shard (tools/pmc_prefill.sh).  Summaries land under gpurun_out/; copy what is to be judged into profiles/r03_*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
OUT=$ROOT/gpurun_out/profile; rm -rf $OUT; mkdir -p $OUT
WARM=20
ARGS="--steps 100 --warmup $WARM --no-cpu-baseline --no-decode"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 < /dev/null
echo "trace rc=$?"; tail -1 $OUT/trace.log | cut -c1-600
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/bench.py $ARGS > $OUT/pmc_$c.log 2>&1 < /dev/null
  echo "pmc $c rc=$?"
done
python $ROOT/tools/profile_summary.py $OUT $WARM | tee $OUT/summary.txt
cd $ROOT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16")
echo "== headline (causal B=16 H=32 S=4096 D=128)"
PMC_OUT=pmc bash tools/pmc_prefill.sh "${SETS[@]}" > gpurun_out/pmc_prefill.log 2>&1
tail -n 40 gpurun_out/pmc_prefill.log
echo "== configs[4] shard (full B=16 H=32 S=8192 D=128)"
SHAPE=16,32,8192 CAUSAL=0 N=3 PMC_OUT=pmc_c5 bash tools/pmc_prefill.sh "${SETS[@]}" > gpurun_out/pmc_prefill_c5.log 2>&1
tail -n 40 gpurun_out/pmc_prefill_c5.log
