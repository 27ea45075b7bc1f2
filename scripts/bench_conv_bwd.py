"""Time the full backward of the paired ResBlock conv (dgrad || wgrad on two streams) with / without the direct reflect dgrad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda")
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda")
for l in (l1, l2):
    l.repack(); l.weight.grad = torch.zeros_like(l.weight); l.bias.grad = torch.zeros_like(l.bias)

def run(nimg, iters=20):
    x = (torch.rand(nimg, 64, 64, 256, device="cuda") * 2 - 1).to(dt).requires_grad_(True)
    dy = (torch.rand(nimg, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    def once():
        y = ops.PairConvFn.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, l1, l2, nimg // 2)
        y.backward(dy)
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): once()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for direct in (False, True):
    for par in (False, True):
        ops.REFLECT_DGRAD_DIRECT, ops.PARALLEL_BACKWARD = direct, par
        print(f"direct_dgrad={direct} parallel_backward={par}: fwd+bwd B16 {run(16):7.1f} us   B8 {run(8):7.1f} us", flush=True)
