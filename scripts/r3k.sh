source scripts/r3_run.sh r3k
step wide 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad"
tail -15 gpurun_out/r3k_wide.log
