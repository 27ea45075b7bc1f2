set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/wgpmc
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python scripts/run_wgrad.py 20 > $O/sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/wgpmc/sq/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'wgrad' in k: print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
