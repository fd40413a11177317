# other shapes with the default (exact-scale, impl 1) and the opt-in prescaled (impl 3) kernels
set -e
python -u tools/prefill_ab.py 1 3 --noncausal
python -u tools/prefill_ab.py 1 3 --noncausal --shape=16,32,8192
python -u tools/prefill_ab.py 1 3 --fp16
python -u tools/prefill_ab.py 1 3 --d64
python -u tools/prefill_ab.py 1 3 --d64 --noncausal
python -u tools/prefill_ab.py -1 --d64 --noncausal --shape=8,16,1024
python -u tools/prefill_ab.py 1 3 --shape=4,32,16384
