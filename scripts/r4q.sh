source scripts/r3_run.sh r4q
step t 600 python -m pytest tests/test_fp8_gpu.py -q -m gpu -k "quantize_multi or quantis"
tail -5 gpurun_out/r4q_t.log
