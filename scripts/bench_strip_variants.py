"""Dominant kernel (ResBlock 3x3 256->256 at 64x64, paired) under the strip-kernel selection hook: forward (with fused IN
statistics) and reflect dgrad (+border, +residual add), batch 16 and 8.  python scripts/bench_strip_variants.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
ref = {}
for mode in (1, 2, 1, 2):
    lib.uig_debug_set_strip(mode)
    for B in (16, 8):
        x = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B)) * 2 - 1).to(dt)
        r = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B + 1)) * 2 - 1).to(dt)
        f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
        g = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
        y, dx = f(), g()
        key = B
        if mode == 1 and key not in ref: ref[key] = (y.clone(), y._uig_in_partial[0].clone(), dx.clone())
        dy = float((y.float() - ref[key][0].float()).abs().max()); dp = float((y._uig_in_partial[0] - ref[key][1]).abs().max()); dd = float((dx.float() - ref[key][2].float()).abs().max())
        tf, tg = t(f), t(g)
        fl = 2.0 * B * 4096 * 256 * 2304 / 1e6
        print(f"strip mode {mode} B{B}: fwd {tf:6.1f} us ({fl/tf:5.0f} TF)  dgrad+border+res {tg:6.1f} us   | vs mode 1: max|dy| {dy:.3g} max|dstats| {dp:.3g} max|ddx| {dd:.3g}", flush=True)
lib.uig_debug_set_strip(1)
