source scripts/r3_run.sh r5e
step t 300 python -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "strip64 or strip128 or infer or translator or Translator"
step inf 300 python scripts/bench_infer.py
tail -3 gpurun_out/r5e_t.log; cat gpurun_out/r5e_inf.log
