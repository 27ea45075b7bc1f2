"""Split one graph-replayed step of gpurun_out/gtrace/kernel_trace.csv into phases (G fwd, G bwd, D fwd, D bwd, Adam) and report
wall time, per-queue busy time and the top kernels of each phase."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/gtrace/kernel_trace.csv')))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_flat_kernel' in r['n']]
t1 = rows[adam[-1]]['e']; t0 = rows[adam[-3]]['e']
win = [r for r in rows if r['s'] >= t0 and r['e'] <= t1]
# phase boundaries: loss kernels. G phase has 6 losses (l1 x4 + mse x2) then backward; D phase 4 mse losses.
loss_idx = [i for i, r in enumerate(win) if 'loss_partial' in r['n']]
print('kernels', len(win), 'step ms', (t1 - t0) / 1e6, 'loss kernels at', loss_idx)
def report(name, ks):
    if not ks: return
    a, b = min(k['s'] for k in ks), max(k['e'] for k in ks)
    byq = collections.Counter(); byn = collections.Counter()
    for k in ks: byq[k['Queue_Id']] += k['e'] - k['s']; byn[k['n']] += k['e'] - k['s']
    print(f"{name:8s} wall {(b - a) / 1e6:7.3f} ms | kernels {len(ks):4d} | busy per queue " + ' '.join(f"q{q}:{v / 1e6:.2f}" for q, v in sorted(byq.items())))
    print('          ' + ' | '.join(f"{n} {v / 1e6:.2f}" for n, v in byn.most_common(6)))
# G forward = up to the first loss; G backward = from last G loss to the pack/first D conv...; use the positions of loss kernels
g_first, g_last = loss_idx[0], loss_idx[5]
d_first, d_last = loss_idx[6], loss_idx[-1]
adam_i = [i for i, r in enumerate(win) if 'adam_flat_kernel' in r['n']]
report('pre+Gfwd', win[:g_first])
report('G bwd', win[g_last + 1:d_first if False else None] if False else [k for k in win[g_last + 1:] if k['s'] < win[d_first]['s'] and win.index(k) < d_first - 40])
report('G bwd..D fwd', win[g_last + 1:d_first])
report('D bwd..', win[d_last + 1:])

# coarse timeline of the whole step: per 0.5 ms bucket, busy fraction of each queue and the dominant kernel
print()
qs = sorted({k['Queue_Id'] for k in win})
bucket = 500_000
nb = int((t1 - t0 + bucket - 1) // bucket)
for b in range(nb):
    a, e = t0 + b * bucket, t0 + (b + 1) * bucket
    line = f"{b * 0.5:5.1f} ms "
    dom = collections.Counter()
    for q in qs:
        busy = sum(max(0, min(k['e'], e) - max(k['s'], a)) for k in win if k['Queue_Id'] == q)
        line += f" q{q}:{100 * busy / bucket:3.0f}%"
    for k in win:
        ov = max(0, min(k['e'], e) - max(k['s'], a))
        if ov: dom[k['n'][:28]] += ov
    line += "  " + ", ".join(f"{n} {v / bucket * 100:.0f}" for n, v in dom.most_common(3))
    print(line)
