set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/inprof
rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
L = u.lib
for B in (8, 16):
    x = (torch.randn(B, 64, 64, 256, device="cuda")).to(torch.bfloat16).requires_grad_(True)
    dy = (torch.randn(B, 64, 64, 256, device="cuda")).to(torch.bfloat16)
    for _ in range(20):
        y = ops.InstNormActFn.apply(x, None, L.ACT_RELU, 0.0, 1e-5)
        y.backward(dy)
        x.grad = None
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $O/run.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/inprof/t/*/*_kernel_trace.csv')[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    by[(n, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(by.items()):
    if 'in_' in k[0]: print(f"{k[0]:42s} grid {k[1]:5d}x{k[2]:3d} n={len(v):3d} avg {sum(v)/len(v):7.1f} us  min {min(v):6.1f}")
PY
