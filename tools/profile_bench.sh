#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of `bench.py`, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) over the same command.
# Output under gpurun_out/profile/; copy the summaries you want judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/profile; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 100 --warmup 20 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 < /dev/null
echo "trace rc=$?"; tail -1 $OUT/trace.log | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_$c.log 2>&1 < /dev/null
  echo "pmc $c rc=$?"
done
python $GRAFT_REPO_ROOT/tools/profile_summary.py $OUT | tee $OUT/summary.txt
