source scripts/r3_run.sh r3g
for d in 16 8 4; do UIG_FIN_DIV=$d step fin$d 300 python scripts/bench_in_fin.py; done
for d in 16 8 4; do echo "--- UIG_FIN_DIV=$d"; cat gpurun_out/r3g_fin$d.log; done
