#!/bin/bash
# PMC passes over the headline prefill kernel (run on the GPU box).  Counters in their own
# runs, no trace domains besides kernel-trace (gpurun refuses other combinations).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/${PMC_OUT:-pmc}; rm -rf $OUT; mkdir -p $OUT     # (SHAPE=B,H,S CAUSAL=0/1 HD IMPL: tools/prefill_once.py)
export IMPL=${IMPL:--1}      # tools/prefill_once.py passes it to sfa_debug_set("prefill_impl", ...)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 < /dev/null
i=0
for set in "${@}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python $GRAFT_REPO_ROOT/tools/prefill_once.py > $OUT/pass$i.log 2>&1 < /dev/null
  echo "pass $i ($set) rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
