set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/wg
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/run_wgrad.py 20 > $O/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python scripts/run_wgrad.py 20 > $O/sq.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -- python scripts/run_wgrad.py 20 > $O/sq2.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
for name in ('sq','sq2'):
    f=glob.glob(f'gpurun_out/wg/{name}/*/*_counter_collection.csv')
    if not f: print(name,'missing'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'wgrad_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(name,k,'%.3g'%(sum(v)/len(v)))
for r in csv.DictReader(open(glob.glob('gpurun_out/wg/trace/*/*_kernel_stats.csv')[0])):
    if 'wgrad' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'])
PY
