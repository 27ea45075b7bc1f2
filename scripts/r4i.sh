source scripts/r3_run.sh r4i
step mem 600 python scripts/dp_memory.py
cat gpurun_out/r4i_mem.log; tail -3 gpurun_out/r4i_mem.err
