set -e
python -u tools/prefill_ab.py 1 10 --noncausal
python -u tools/prefill_ab.py 1 10 --noncausal --shape=16,32,8192
python -u tools/prefill_ab.py 1 10 --fp16
python -u tools/prefill_ab.py 1 10 --d64
python -u tools/prefill_ab.py 1 10 --d64 --noncausal
python -u tools/prefill_ab.py 1 20 21 --d64 --noncausal --shape=8,16,1024
python -u tools/prefill_ab.py 1 20 21 --shape=1,32,4096
python -u tools/prefill_ab.py 1 --shape=4,32,16384
