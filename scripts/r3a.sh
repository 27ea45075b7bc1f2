source scripts/r3_run.sh r3a
step tests 1100 python -m pytest tests -q -m gpu
step bench 600 python bench.py
step pg 600 python tests/_pg_lifecycle_worker.py
tail -5 gpurun_out/r3a_tests.log; cat gpurun_out/r3a_bench.log | head -c 3000; tail -5 gpurun_out/r3a_pg.log
