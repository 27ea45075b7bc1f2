source scripts/r3_run.sh r4t
step t64 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "strip64 or strip128"
step inf 600 python scripts/bench_infer_small.py
tail -8 gpurun_out/r4t_t64.log; cat gpurun_out/r4t_inf.log
