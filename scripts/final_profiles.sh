# Round-end evidence: (1) rocprofv3 kernel trace + stats of the bench (eager so every kernel is attributed),
# (2) kernel trace + PMC passes of the dominant kernel.  Summaries are copied into profiles/ by the caller.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UIG_BENCH_SOFT_EXIT=1
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $O/step.log 2>&1
cp $O/step/*/*_kernel_stats.csv $O/step_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/run_dominant.py 30 > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python scripts/run_dominant.py 30 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python scripts/run_dominant.py 30 > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python scripts/run_dominant.py 30 > $O/sq.log 2>&1 || true
cp $O/trace/*/*_kernel_stats.csv $O/dominant_kernel_stats.csv
rm -rf $O/step
tail -1 $O/step.log | cut -c1-200
