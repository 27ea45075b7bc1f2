# rocprofv3 passes for the dominant kernel: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/dom
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/run_dominant.py 30 > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python scripts/run_dominant.py 30 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python scripts/run_dominant.py 30 > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python scripts/run_dominant.py 30 > $O/sq.log 2>&1 || true
find $O -name "*.csv" | head -20
