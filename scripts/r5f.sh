source scripts/r3_run.sh r5f
step small 600 python scripts/bench_infer_small.py
cat gpurun_out/r5f_small.log
