"""Would one 24-image weight-gradient launch (both generator passes) beat 16 + 8 images for the layers on the GENERIC split-K
kernel and the 7x7 head kernel?  Partial + reduce, paired networks, bf16, one HIP graph per case.
python scripts/bench_wgrad_combine_generic.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16


def graph_time(fn, n=10, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                fn()
        g.replay(); s.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return statistics.median(ts)


layers = [("down1 3x3s2 64->128 @256", "conv", 64, 128, 3, 2, 1, "zero", 256), ("down2 3x3s2 128->256 @128", "conv", 128, 256, 3, 2, 1, "zero", 128),
          ("up1 convT 256->128 @64", "convT", 256, 128, 3, 2, 1, "zero", 64), ("up2 convT 128->64 @128", "convT", 128, 64, 3, 2, 1, "zero", 128),
          ("stem 7x7 3->64 @256", "conv", 3, 64, 7, 1, 3, "reflect", 256), ("head 7x7 64->3 @256", "conv", 64, 3, 7, 1, 3, "reflect", 256)]
for name, kind, ci, co, k, st, pd, pm, hw in layers:
    l1 = networks.ConvLayer(kind, ci, co, k, st, pd, pm, dtype=dt, device="cuda"); l1.repack()
    l2 = networks.ConvLayer(kind, ci, co, k, st, pd, pm, dtype=dt, device="cuda"); l2.repack()
    for l in (l1, l2):
        l.weight.grad = torch.zeros_like(l.weight); l.bias.grad = torch.zeros_like(l.bias)
    res = {}
    for B in (16, 8, 24):
        x = (torch.rand(B, hw, hw, l1.spec.cin_p, device="cuda") * 2 - 1).to(dt)
        y = ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias)
        dy = torch.randn_like(y)
        def f(x=x, dy=dy, B=B):
            pp = ops.conv_wgrad_pair_partial(l1.spec, x, dy, B // 2)
            assert ops._param_grads_pair((l1, l2), l1.spec, x, dy, B // 2, None, pp, bias=False)
        res[B] = graph_time(f)
    print(f"{name:28s} 16 images {res[16]:6.1f} + 8 images {res[8]:6.1f} = {res[16] + res[8]:6.1f} us | 24 in one launch {res[24]:6.1f} us   saves {res[16] + res[8] - res[24]:5.1f} us", flush=True)
