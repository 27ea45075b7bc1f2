source scripts/r3_run.sh r3v
step t128 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "strip128 or reflect_dgrad_border or conv_fwd_bwd"
step inf 600 python scripts/bench_infer_stages.py
step tinf 600 python -m pytest tests/test_inference_gpu.py -q -m gpu
tail -5 gpurun_out/r3v_t128.log; cat gpurun_out/r3v_inf.log; tail -3 gpurun_out/r3v_tinf.log
