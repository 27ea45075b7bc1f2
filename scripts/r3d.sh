source scripts/r3_run.sh r3d
step normconv 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "norm_conv"
step tests 1100 python -m pytest tests -q -m gpu
step bench 600 python bench.py
UIG_NORM_CONV=0 step bench0 600 python bench.py --no-cpu-baseline --no-other-configs
tail -8 gpurun_out/r3d_normconv.log; tail -8 gpurun_out/r3d_tests.log; cat gpurun_out/r3d_bench.log | head -c 3000; echo; cat gpurun_out/r3d_bench0.log | head -c 1800
