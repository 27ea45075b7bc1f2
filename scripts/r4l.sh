source scripts/r3_run.sh r4l
step ts2 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "stride2_matches"
step twg 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad"
tail -15 gpurun_out/r4l_ts2.log; tail -3 gpurun_out/r4l_twg.log
