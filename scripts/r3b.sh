source scripts/r3_run.sh r3b
step two 900 python -m pytest tests/test_fp8_gpu.py tests/test_model_gpu.py -q -m gpu -k "config5 or config2"
export UIG_TEST_PG_INPROCESS=1
step inproc 1100 python -m pytest tests -q -m gpu -x
grep -E "fake_B vs|weight gradient|passed|failed" gpurun_out/r3b_two.log | head; tail -5 gpurun_out/r3b_inproc.log
