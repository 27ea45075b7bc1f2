source scripts/r3_run.sh r3c
step tests 1150 python -m pytest tests -q -m gpu
step bench 600 python bench.py
step benchin 300 python scripts/bench_in.py
export UIG_TEST_PG_INPROCESS=1
step inproc 1100 python -m pytest tests -q -m gpu --deselect tests/test_fp8_gpu.py::test_train_step_config5_b8_256_fp8_graph_vs_same_rounding_oracle --deselect tests/test_model_gpu.py::test_train_step_config2_b4_256_bf16_graph_vs_oracle
tail -8 gpurun_out/r3c_tests.log; cat gpurun_out/r3c_bench.log | head -c 3500; cat gpurun_out/r3c_benchin.log; tail -5 gpurun_out/r3c_inproc.log
