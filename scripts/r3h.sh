source scripts/r3_run.sh r3h
# hypothesis for the round-2 replay segfault: graphs with parallel branches (side stream for the parameter gradients, update
# stream: the defaults at the time) captured AFTER a process group was destroyed in the same process.  ONE run, under the
# debugger so that a native fault leaves its frames.
export UIG_PARALLEL_BACKWARD=1 UIG_OVERLAP_UPDATE=1
step gdb 420 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex run -ex bt -ex "info threads" --args python tests/_pg_lifecycle_worker.py
tail -40 gpurun_out/r3h_gdb.log; tail -5 gpurun_out/r3h_gdb.err
