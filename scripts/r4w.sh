source scripts/r3_run.sh r4w
step t 900 python -m pytest tests/test_pipeline_gpu.py tests/test_ops_gpu.py -q -m gpu -k "fused_instnorm or translator or Translator or infer or strip64 or strip128 or hypothesis"
step inf 600 python scripts/bench_infer.py
step small 600 python scripts/bench_infer_small.py
tail -4 gpurun_out/r4w_t.log; cat gpurun_out/r4w_inf.log; cat gpurun_out/r4w_small.log
