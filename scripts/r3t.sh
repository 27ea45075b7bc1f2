source scripts/r3_run.sh r3t
step fp8t 900 python -m pytest tests/test_fp8_gpu.py -q -m gpu -x
step stamp 300 python scripts/stamp_fp8.py 32
step stamp16 300 python scripts/stamp_fp8.py 16
tail -3 gpurun_out/r3t_fp8t.log; cat gpurun_out/r3t_stamp.log gpurun_out/r3t_stamp16.log; tail -5 gpurun_out/r3t_stamp.err
