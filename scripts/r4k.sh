source scripts/r3_run.sh r4k
step t512 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "config4_b2_512 or two_rank or rccl or lifecycle"
step st256 300 python scripts/stamp_strip_pk.py 16
step st128 300 env STAMP_GRID=128 python scripts/stamp_strip_pk.py 16
step st64 300 env STAMP_GRID=64 python scripts/stamp_strip_pk.py 16
tail -4 gpurun_out/r4k_t512.log
for f in st256 st128 st64; do echo "== $f"; head -12 gpurun_out/r4k_$f.log; done
