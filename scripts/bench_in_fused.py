"""InstanceNorm backward: three launches (statistics, finalize, apply) vs the fused one-launch kernel, at the ResBlock map of the
benchmark (16 and 8 images of 64x64x256) and the PatchGAN maps.  python scripts/bench_in_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
lib = u.lib.lib()
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for shape in ((16, 64, 64, 256), (8, 64, 64, 256), (16, 64, 64, 128), (16, 32, 32, 256)):
    B, H, W, C = shape
    x = torch.randn(shape, device="cuda").to(torch.bfloat16)
    dy = torch.randn(shape, device="cuda").to(torch.bfloat16)
    stats = torch.stack([x.float().mean((1, 2)), torch.rsqrt(x.float().var((1, 2), unbiased=False) + 1e-5)], -1).contiguous()
    res = {}
    for rnd in range(3):
        for fused in (False, True):
            ops.FUSED_IN_BWD = fused
            res.setdefault(fused, []).append(t(lambda: ops.instnorm_backward(dy, x, stats, u.lib.ACT_RELU, 0.0)))
    ops.check_sync_errors("cuda")
    mb = B * H * W * C * 2 / 1e6
    for fused in (False, True):
        v = sorted(res[fused])[1]
        print(f"{shape}: {'fused' if fused else 'three launches'}: {v:7.2f} us  ({(3 if fused else 5) * mb / v / 1e6 * 1e6 / 1e3:.2f} TB/s algorithmic)  {['%.1f' % q for q in res[fused]]}")
