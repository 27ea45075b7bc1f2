"""Is the strip convolution slower in the step than back to back because it starts on cold caches?  The paired 16-image ResBlock launch
timed (HIP events around each launch) (a) back to back, (b) behind an InstanceNorm apply pass over a tensor of its input's size (what
precedes it in the step), (c) behind a 512-MB fill (L2 and Infinity Cache evicted).  python scripts/bench_cold.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
junk = torch.empty(128 << 20, device="cuda", dtype=torch.float32)
for B in (16, 8):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    z = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    conv = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
    def norm():
        with torch.no_grad():
            return ops.InstNormActFn.apply(z, None, u.lib.ACT_RELU, 0.0, 1e-5)
    for name, pre in (("back to back", None), ("behind an InstanceNorm pass", norm), ("behind a 512-MB fill", lambda: junk.fill_(1.0))):
        for _ in range(5):
            if pre: pre()
            conv()
        evs = []
        for _ in range(40):
            if pre: pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); conv(); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        print(f"B{B:2d} conv fwd {name:30s}: median {ts[len(ts)//2]:6.1f} us  (min {ts[0]:.1f}, p90 {ts[int(len(ts)*0.9)]:.1f})")
