source scripts/r3_run.sh r4y
step t 900 python -m pytest tests/test_model_gpu.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py -q -m gpu -k "two_rank or infer or translator or Translator or strip64 or hypothesis"
step small 600 python scripts/bench_infer_small.py
tail -4 gpurun_out/r4y_t.log; cat gpurun_out/r4y_small.log
