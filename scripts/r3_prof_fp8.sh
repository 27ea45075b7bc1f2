# rocprofv3 passes of the MX fp8 strip kernel at configs[4]'s launch size (32 images): --stats, FETCH_SIZE / WRITE_SIZE, SQ counters
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r3p}
D=gpurun_out/${T}_fp8
rm -rf $D; mkdir -p $D
SQ1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python scripts/run_dominant.py 2000 fp8 > $D/trace.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python scripts/run_dominant.py 30 fp8 > $D/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python scripts/run_dominant.py 30 fp8 > $D/write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $D/sq -- python scripts/run_dominant.py 30 fp8 > $D/sq.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $D/sq2 -- python scripts/run_dominant.py 30 fp8 > $D/sq2.log 2>&1
python scripts/pmc_summary.py $D gpurun_out/${T}_fp8_kernel_pmc.json
cp $D/trace/*/*_kernel_stats.csv gpurun_out/${T}_fp8_kernel_stats.csv
rm -rf $D
