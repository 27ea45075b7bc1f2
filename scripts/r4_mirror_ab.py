import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib(); dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
def t(fn, n=60):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
res = {}
for rnd in range(3):
    for B in (16, 8):
        x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
        r = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
        f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
        g = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
        g0 = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2))
        for dm in (5, 0):
            lib.uig_debug_set_strip_pk(dm, 0)
            res.setdefault((B, dm, "fwd"), []).append(t(f))
            for m in (1, 3, 7):
                lib.uig_debug_set_mirror(m)
                res.setdefault((B, dm, f"mirror{m}+res"), []).append(t(g))
                res.setdefault((B, dm, f"mirror{m}"), []).append(t(g0))
            lib.uig_debug_set_mirror(1)
lib.uig_debug_set_strip_pk(0, 0)
for k, v in sorted(res.items()): print(k, f"{sorted(v)[1]:.1f}")
