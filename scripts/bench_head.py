"""7x7 head forward (64 -> 3, tanh) per kernel-selection hook.  python scripts/bench_head.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib(); dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 64, 3, 7, 1, 3, "reflect", act=u.lib.ACT_TANH, dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 64, 3, 7, 1, 3, "reflect", act=u.lib.ACT_TANH, dtype=dt, device="cuda"); l2.repack()
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for mode in (2, 1, 2, 1):
    lib.uig_debug_set_rowstrip(mode)
    for B in (16, 8):
        x = (torch.rand(B, 256, 256, 64, device="cuda") * 2 - 1).to(dt)
        print(f"rowstrip mode {mode} B{B}: {t(lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, (l2.wp_fwd, l2.bias, B // 2))):7.1f} us", flush=True)
lib.uig_debug_set_rowstrip(1)
# stem input gradient: dy (B,256,256,64) -> padded gradient (B,262,262,8) -> fold
st = networks.ConvLayer("conv", 3, 64, 7, 1, 3, "reflect", dtype=dt, device="cuda"); st.repack()
st2 = networks.ConvLayer("conv", 3, 64, 7, 1, 3, "reflect", dtype=dt, device="cuda"); st2.repack()
for mode in (2, 1, 2, 1):
    lib.uig_debug_set_rowstrip(mode)
    for B in (8,):
        dy = (torch.rand(B, 256, 256, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(5)) * 2 - 1).to(dt)
        f = lambda: ops.conv_dgrad(st.spec, dy, st.wp_dgrad, (256, 256), (st2.wp_dgrad, None, B // 2))
        out = f()
        if mode == 2: ref = out.clone()
        print(f"rowstrip mode {mode} stem dgrad B{B}: {t(f):7.1f} us   max|d| vs mode 2: {float((out.float() - ref.float()).abs().max()):.3g} (scale {float(ref.float().abs().max()):.3g})", flush=True)
lib.uig_debug_set_rowstrip(1)
