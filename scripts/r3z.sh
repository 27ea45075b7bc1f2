source scripts/r3_run.sh r3z
step tests 1100 python -m pytest tests -q -m gpu
tail -6 gpurun_out/r3z_tests.log
