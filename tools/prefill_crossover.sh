#!/bin/bash
# Where the 4-wave persistent kernel (impl 40) overtakes the 8-wave 256-row (1) and the 128-row (20) kernels:
# the data behind the auto rule in csrc/prefill_dispatch.hip.  Run on the GPU box.  The last block: query count != key
# count (few keys per row: the 4-wave kernel's per-q-tile costs against very short rows).
cd "${GRAFT_REPO_ROOT:-.}"
for shape in 1,8,4096 1,16,4096 1,32,4096 2,32,4096 4,32,4096 2,32,2048 8,32,2048 16,32,2048 16,32,1024 32,32,1024 4,32,8192 1,32,16384; do
  for mode in "" "--noncausal"; do
    timeout -k 10 120 python -u tools/prefill_ab.py 1 20 40 --shape=$shape $mode 2>&1 | grep impl
  done
done
for sk in 64 128 256 512 1024; do
  timeout -k 10 120 python -u tools/prefill_ab.py 1 20 40 --shape=8,32,4096 --sk=$sk --noncausal 2>&1 | grep impl
done
for sk in 1024 2048 8192; do
  timeout -k 10 120 python -u tools/prefill_ab.py 1 20 40 --shape=8,32,4096 --sk=$sk 2>&1 | grep impl
done
