# kernel trace of the graph step with every kernel serialised on one stream (no side-stream overlap): each duration is
# the kernel's own cost.  Output: gpurun_out/strace/kernel_trace.csv
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UIG_BENCH_SOFT_EXIT=1 UIG_PARALLEL_BACKWARD=0 UIG_OVERLAP_UPDATE=0
O=gpurun_out/strace
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv
rm -rf $O/t
tail -2 $O/log.txt
