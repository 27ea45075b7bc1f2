"""Would ONE 24-image weight-gradient launch per ResBlock conv (both generator passes of the step together) beat the two launches
(16 + 8 images) the step issues today?  Partial + reduce, paired networks, bf16.  python scripts/bench_wgrad_combine.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
for l in (l1, l2):
    l.weight.grad = torch.zeros_like(l.weight); l.bias.grad = torch.zeros_like(l.bias)
def t(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
def run(B):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.randn(B, 64, 64, 256, device="cuda") * 0.5).to(dt)
    def f():
        pp = ops.conv_wgrad_pair_partial(l1.spec, x, dy, B // 2)
        assert ops._param_grads_pair((l1, l2), l1.spec, x, dy, B // 2, None, pp)
    return t(f)
for rnd in range(3):
    a, b, c = run(16), run(8), run(24)
    print(f"round {rnd}: 16 images {a:6.1f} us + 8 images {b:6.1f} us = {a + b:6.1f} us   |   24 images in one launch {c:6.1f} us   (partial + reduce + bias grad)", flush=True)
