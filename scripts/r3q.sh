source scripts/r3_run.sh r3q
step stamp 300 python scripts/stamp_fp8.py 32
step stamp16 300 python scripts/stamp_fp8.py 16
cat gpurun_out/r3q_stamp.log gpurun_out/r3q_stamp16.log; tail -5 gpurun_out/r3q_stamp.err
