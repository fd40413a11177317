# 256-row paired kernel (impl 1) vs 128-row kernel (impl 20) on grids that cannot fill 256 CUs
set -e
for shp in 1,8,4096 1,16,4096 1,16,2048 1,32,2048 2,32,1024 4,32,1024 1,32,1024 8,32,512 2,8,8192 1,24,4096; do
  python -u tools/prefill_ab.py 1 20 --shape=$shp
done
python -u tools/prefill_ab.py 1 20 --shape=1,16,4096 --noncausal
python -u tools/prefill_ab.py 1 20 --shape=2,32,1024 --noncausal
python -u tools/prefill_ab.py 1 20 --shape=1,24,4096 --noncausal
