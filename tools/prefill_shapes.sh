#!/bin/bash
# Shape sweep of the library's own choice (prefill_impl -1) beside the opt-in prescaled flavour where it exists (41: the 4-wave
# kernel forced; 3: the 8-wave one): sequence lengths 1k..32k, both masks, fp16, head_dim 64 / 256.  Run on the GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
run() { timeout -k 10 180 python -u tools/prefill_ab.py "$@" 2>&1 | grep impl; }
for shape in 64,32,1024 32,32,2048 16,32,4096 8,32,8192 4,32,16384 2,32,32768; do
  run -1 41 --shape=$shape
  run -1 41 --shape=$shape --noncausal
done
run -1 41 --fp16
run -1 41 --fp16 --noncausal --shape=16,32,8192
run -1 3 --d64 --shape=8,16,1024 --noncausal
run -1 3 --d64
run -1 3 --d64 --noncausal
run -1 --d256 --shape=8,16,4096
run -1 --d256 --shape=8,16,4096 --noncausal
