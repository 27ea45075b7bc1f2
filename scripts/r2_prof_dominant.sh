# rocprofv3 passes for the dominant kernel (paired ResBlock conv launch, 16 images): kernel trace + stats, FETCH_SIZE and
# WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md: HBM), SQ counters; then the SQ counters again for the XOR-swizzle
# variant (UIG_DEBUG_HOOKS=strip_pk=2:0) and the one-tile-per-block kernel (strip=3).  bash scripts/r2_prof_dominant.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/dom2
rm -rf $O; mkdir -p $O
SQ="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/run_dominant.py 3000 > $O/trace.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python scripts/run_dominant.py 30 > $O/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python scripts/run_dominant.py 30 > $O/write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq -- python scripts/run_dominant.py 30 > $O/sq.log 2>&1
python scripts/pmc_summary.py $O gpurun_out/r02_dominant_pmc.json > /dev/null
cp $O/trace/*/*_kernel_stats.csv gpurun_out/r02_dominant_kernel_stats.csv
for v in "strip_pk=2:0" "strip=3"; do
  T=$O/sq_$(echo $v | tr '=:' '__'); mkdir -p $T
  UIG_DEBUG_HOOKS=$v timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $T/sq -- python scripts/run_dominant.py 30 > $T/log.txt 2>&1
  python scripts/pmc_summary.py $T gpurun_out/r02_dominant_pmc_$(echo $v | tr '=:' '__').json > /dev/null
done
rm -rf $O
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_dominant_pmc*.json")):
    j = json.load(open(f))
    print(f, j.get("kernel"), "avg_us", j.get("avg_duration_us"), "conflict/active", round(j.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, j.get("SQ_LDS_IDX_ACTIVE", 1)), 4),
          "wait_inst/wave", round(j.get("SQ_WAIT_INST_ANY", 0) / max(1, j.get("SQ_WAVE_CYCLES", 1)), 3), "hbm", j.get("hbm_bytes_per_launch"))
PY
