import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib; lib = L.lib()
dt = torch.bfloat16
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
lr = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); lr.repack()
lz = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "zero", dtype=dt, device="cuda"); lz.repack()
for B in (8, 16):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    bord = torch.empty(B, 8, 64, 256, device="cuda", dtype=dt)
    s = torch.cuda.current_stream().cuda_stream
    print(f"B{B}: fwd reflect {t(lambda: ops.conv_forward(lr.spec, x, lr.wp_fwd, lr.bias)):.1f} us | fwd zero {t(lambda: ops.conv_forward(lz.spec, x, lz.wp_fwd, lz.bias)):.1f} us"
          f" | dgrad reflect {t(lambda: ops.conv_dgrad(lr.spec, dy, lr.wp_dgrad, (64, 64))):.1f} us | dgrad zero {t(lambda: ops.conv_dgrad(lz.spec, dy, lz.wp_dgrad, (64, 64))):.1f} us"
          f" | border only {t(lambda: L.check(lib.uig_reflect3x3_dgrad_border(dy.data_ptr(), lr.wp_dgrad.data_ptr(), None, 0, bord.data_ptr(), B, 64, 64, 256, 256, 256, 1, s), 'b')):.1f} us")
