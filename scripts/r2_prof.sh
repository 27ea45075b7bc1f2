# round-2 measurement pass on one box: bench line (exit code checked, 2 runs + one with the RCCL path forced), serialised kernel
# trace of the step, rocprofv3 --stats of the default bench.  Outputs under gpurun_out/r2p_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r2p}
for i in 1 2; do timeout -k 10 400 python bench.py > gpurun_out/${T}_bench$i.json 2> gpurun_out/${T}_bench$i.err; echo "bench run $i exit code $?" | tee -a gpurun_out/${T}_exit.log; done
RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 timeout -k 10 400 python bench.py --force-comm --no-cpu-baseline --steps 10 > gpurun_out/${T}_bench_fc.json 2> gpurun_out/${T}_bench_fc.err; echo "bench --force-comm exit code $?" | tee -a gpurun_out/${T}_exit.log
# the default bench.py command itself under rocprofv3 --kernel-trace --stats: the stats summary, and the roofline region's launches
# picked out of the trace and set beside the live number of the same process (scripts/roofline_region_from_trace.py)
B=gpurun_out/${T}_benchprof
rm -rf $B; mkdir -p $B
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $B/t -- python bench.py --no-cpu-baseline > $B/bench_line.json 2> $B/bench.err
echo "bench.py under rocprofv3 exit code $?" | tee -a gpurun_out/${T}_exit.log
cp $B/t/*/*_kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv
python scripts/roofline_region_from_trace.py $B/t/*/*_kernel_trace.csv $B/bench_line.json gpurun_out/${T}_bench_roofline_region.json > /dev/null 2>> gpurun_out/${T}_exit.log
cp $B/bench_line.json gpurun_out/${T}_bench_line_under_rocprof.json
rm -rf $B
export UIG_PARALLEL_BACKWARD=0 UIG_OVERLAP_UPDATE=0
O=gpurun_out/${T}_strace
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
echo "serial trace exit code $?" | tee -a gpurun_out/${T}_exit.log
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
python scripts/summarize_trace.py $O/kernel_trace.csv gpurun_out/${T}_step_serial_kernels.csv > gpurun_out/${T}_step_serial.txt 2>&1
rm -f $O/kernel_trace.csv
unset UIG_PARALLEL_BACKWARD UIG_OVERLAP_UPDATE
tail -3 gpurun_out/${T}_bench1.json | cut -c1-1500
head -40 gpurun_out/${T}_step_serial.txt
