source scripts/r3_run.sh r3e
step fin 300 python scripts/bench_in_fin.py
cat gpurun_out/r3e_fin.log; tail -3 gpurun_out/r3e_fin.err
