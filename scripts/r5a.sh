source scripts/r3_run.sh r5a
step t 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "strip64 or strip128"
step inf2 300 python scripts/bench_infer.py
step inf4 300 env UIG_DEBUG_HOOKS=strip_stages=4 python scripts/bench_infer.py
tail -3 gpurun_out/r5a_t.log; echo "--- 2 stages"; grep "B=1 256\|B=2 256" gpurun_out/r5a_inf2.log; echo "--- 4 stages"; grep "B=1 256\|B=2 256" gpurun_out/r5a_inf4.log
