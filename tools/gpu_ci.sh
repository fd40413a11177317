#!/bin/bash
# Run on the GPU box via gpurun: smoke -> pytest -m gpu -> short bench.  Each step is bounded by
# its own timeout; a step that times out or is killed stops the sequence (no GPU work after a hang).
# Test/assertion failures do not stop later steps (their logs are what we want back).
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
run_step() {  # name seconds cmd...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a gpurun_out/ci.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1 < /dev/null
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/ci.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!! $name timed out/killed: stopping"; exit $rc; fi
  return 0
}
: > gpurun_out/ci.log
STEPS=${STEPS:-smoke tests bench}
for s in $STEPS; do
  case $s in
    smoke) run_step smoke 300 python __graft_entry__.py smoke ;;
    tests) run_step tests 900 python -u -m pytest tests -m gpu -x -q --timeout 120 ;;
    testsall) run_step tests 900 python -u -m pytest tests -m gpu -q --timeout 120 ;;
    bench) run_step bench 600 python bench.py ;;
    benchfast) run_step bench 300 python bench.py --no-cpu-baseline --no-decode ;;
    prefill) run_step prefill 300 python -u -m pytest tests/test_prefill_gpu.py -m gpu -x -q --timeout 120 ;;
    decode) run_step decode 300 python -u -m pytest tests/test_decode_gpu.py -m gpu -x -q --timeout 120 ;;
    ab) run_step ab 300 python -u tools/prefill_ab.py ${AB_ARGS:-1 40} ;;
    w4) run_step w4 600 python -u -m pytest tests/test_prefill_gpu.py -m gpu -x -q --timeout 120 -k "w4 or flavours" ;;
    full) run_step full 900 python -u -m pytest tests/test_full_configs_gpu.py -m gpu -x -q --timeout 600 ;;
    variants) run_step variants 900 env SFA_LIB_PATH=$PWD/starflashattention_amd/lib/libStarFlashAttention_ab.so python -u -m pytest tests -m variants -x -q --timeout 120 ;;
    cmd) run_step cmd ${CMD_TIMEOUT:-300} bash -c "$CMD" < /dev/null ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
