"""A/B of two library builds (UIG_LIB_PATH) on the paired ResBlock input-gradient launch: run once per build, alternate outside."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
def t(fn, n=200):
    for _ in range(100): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for B in (16, 8):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    r = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    xin = (torch.randn(B, 64, 64, 256, device="cuda")).to(dt)
    st = torch.rand(B, 256, 2, device="cuda") + 0.5
    plain = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
    fused = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r, bst=(xin, st, 1, 0.0))
    print(f"{os.environ.get('UIG_LIB_PATH', 'default')[-14:]} B{B}: dgrad+border+res {t(plain):6.1f} us   + bwd statistics {t(fused):6.1f} us", flush=True)
