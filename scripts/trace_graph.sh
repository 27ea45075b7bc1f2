set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export UIG_BENCH_SOFT_EXIT=1
O=gpurun_out/gtrace
rm -rf $O; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
cp $O/t/*/*_kernel_trace.csv $O/kernel_trace.csv
rm -rf $O/t
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/gtrace/kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last ~ step: find the last adam kernels; use the window between the 2nd-last and last "adam_flat_kernel" pair ends
adam = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['Kernel_Name']]
# each step has 2 adam kernels (+2 tick); take end of adam #-5.. as step boundary: steps end with adam D
ends = [int(rows[i]['End_Timestamp']) for i in adam]
# last step window: from the end of the adam before the last two, to the end of the last
t1 = ends[-1]; t0 = ends[-3]
win = [r for r in rows if t0 <= int(r['Start_Timestamp']) and int(r['End_Timestamp']) <= t1]
print('kernels in window', len(win), 'window ms', (t1 - t0) / 1e6)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in win)
print('sum of kernel durations ms', busy / 1e6)
# union coverage
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
cov = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: cov += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
cov += ce - cs
print('time with >=1 kernel running ms', cov / 1e6, ' idle ms', (t1 - t0 - cov) / 1e6)
by = collections.Counter()
for r in win: by[r['Kernel_Name'].split('(')[0][:60]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in by.most_common(14): print(f'{v/1e6:8.3f} ms  {k}')
PY
