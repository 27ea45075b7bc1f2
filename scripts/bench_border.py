import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib; lib = L.lib()
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); layer.repack()
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for tile in (0, 3, 64):
    lib.uig_debug_set_tile(tile if tile else 0)
    for B in (8, 16):
        dy = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
        bord = torch.empty(B, 8, 64, 256, device="cuda", dtype=dt)
        f = lambda: L.check(lib.uig_reflect3x3_dgrad_border(dy.data_ptr(), layer.wp_dgrad.data_ptr(), None, 0, bord.data_ptr(), B, 64, 64, 256, 256, 256, 1, torch.cuda.current_stream().cuda_stream), "b")
        print(f"force_tile={tile} border GEMM B{B}: {t(f):.1f} us")
    x = (torch.rand(8, 32, 32, 256, device="cuda") * 2 - 1).to(dt)
    l4 = networks.ConvLayer("conv", 256, 512, 4, 1, 1, "zero", dtype=dt, device="cuda"); l4.repack()
    print(f"force_tile={tile} D4 fwd B8: {t(lambda: ops.conv_forward(l4.spec, x, l4.wp_fwd, l4.bias)):.1f} us")
