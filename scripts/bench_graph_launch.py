"""GPU-side launch-to-launch time of the ResBlock strip conv inside a HIP graph (no host launch cost), batch 8 and 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); layer.repack()
for B in (8, 16):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    for want_stats in (False, True):
        f = lambda: ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias, want_in_stats=want_stats)
        for _ in range(3): f()
        t0 = time.time(); 
        for _ in range(200): f()
        t_host = (time.time() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            f()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): keep = f()
        torch.cuda.synchronize()
        for _ in range(3): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); e1.synchronize()
        print(f"B{B} stats={want_stats}: graph launch-to-launch {e0.elapsed_time(e1) * 1e3 / 200:.1f} us | host issue cost {t_host:.1f} us/call")
