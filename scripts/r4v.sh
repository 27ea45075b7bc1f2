source scripts/r3_run.sh r4v
step t 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "fused_instnorm or translator or Translator or infer"
step inf 600 python scripts/bench_infer.py
step inf0 600 env UIG_DEBUG_HOOKS=infer_cs=0 python scripts/bench_infer.py
tail -4 gpurun_out/r4v_t.log; echo "--- channel-sliced"; cat gpurun_out/r4v_inf.log; echo "--- whole-C (round 2)"; cat gpurun_out/r4v_inf0.log
