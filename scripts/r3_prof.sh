# round-3 measurement pass on one box (bash scripts/r3_prof.sh [TAG]): bench line x2 with exit codes, bench.py itself under
# rocprofv3 --kernel-trace --stats (+ the roofline region picked out of its trace), the serialised kernel trace of the step, and
# the dominant launch: --stats, FETCH_SIZE / WRITE_SIZE in separate PMC passes, two SQ counter passes (the second one with the
# instruction-mix counters VERDICT round 2 asked for).  Outputs under gpurun_out/${TAG}_*; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r3p}
for i in 1 2; do timeout -k 10 400 python bench.py > gpurun_out/${T}_bench$i.json 2> gpurun_out/${T}_bench$i.err; echo "bench run $i exit code $?" | tee -a gpurun_out/${T}_exit.log; done
B=gpurun_out/${T}_benchprof
rm -rf $B; mkdir -p $B
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $B/t -- python bench.py --no-cpu-baseline --no-other-configs > $B/bench_line.json 2> $B/bench.err
echo "bench.py under rocprofv3 exit code $?" | tee -a gpurun_out/${T}_exit.log
cp $B/t/*/*_kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv
python scripts/roofline_region_from_trace.py $B/t/*/*_kernel_trace.csv $B/bench_line.json gpurun_out/${T}_bench_roofline_region.json > /dev/null 2>> gpurun_out/${T}_exit.log
cp $B/bench_line.json gpurun_out/${T}_bench_line_under_rocprof.json
rm -rf $B
O=gpurun_out/${T}_strace
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/log.txt 2>&1
echo "serial trace exit code $?" | tee -a gpurun_out/${T}_exit.log
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv && rm -rf $O/t
python scripts/summarize_trace.py $O/kernel_trace.csv gpurun_out/${T}_step_serial_kernels.csv gpurun_out/${T}_step_small_layers.csv > gpurun_out/${T}_step_serial.txt 2>&1
rm -f $O/kernel_trace.csv
# ---- dominant launch (as the step runs it: with the InstanceNorm statistics requested)
D=gpurun_out/${T}_dom
rm -rf $D; mkdir -p $D
SQ1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python scripts/run_dominant.py 3000 stats > $D/trace.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python scripts/run_dominant.py 30 stats > $D/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python scripts/run_dominant.py 30 stats > $D/write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $D/sq -- python scripts/run_dominant.py 30 stats > $D/sq.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $D/sq2 -- python scripts/run_dominant.py 30 stats > $D/sq2.log 2>&1
python scripts/pmc_summary.py $D gpurun_out/${T}_dominant_pmc.json > /dev/null
cp $D/trace/*/*_kernel_stats.csv gpurun_out/${T}_dominant_kernel_stats.csv
rm -rf $D
head -c 1500 gpurun_out/${T}_bench1.json; echo
head -45 gpurun_out/${T}_step_serial.txt
cat gpurun_out/${T}_dominant_pmc.json
