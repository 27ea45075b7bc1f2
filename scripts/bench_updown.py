"""Stride-2 / transposed generator layers (down1, down2, up1, up2), forward and input gradient, per tile-selection hook.
  python scripts/bench_updown.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib; lib = L.lib()
dt = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
layers = [("down1", "conv", 64, 128, 256), ("down2", "conv", 128, 256, 128), ("up1", "convT", 256, 128, 64), ("up2", "convT", 128, 64, 128)]
for tile in (0, 1):
    lib.uig_debug_set_tr2(tile)
    for name, kind, ci, co, hw in layers:
        l = networks.ConvLayer(kind, ci, co, 3, 2, 1, dtype=dt, device="cuda"); l.repack()
        x = (torch.rand(B, hw, hw, ci, device="cuda") * 2 - 1).to(dt)
        y = ops.conv_forward(l.spec, x, l.wp_fwd, l.bias)
        dy = torch.rand_like(y)
        gf = 2.0 * B * y.shape[1] * y.shape[2] * co * ci * 9 / (4 if kind == "convT" else 1) / 1e9
        tf = t(lambda: ops.conv_forward(l.spec, x, l.wp_fwd, l.bias)); tb = t(lambda: ops.conv_dgrad(l.spec, dy, l.wp_dgrad, (hw, hw)))
        print(f"tr2={tile:3d} {name:6s} B{B}: fwd {tf:7.1f} us ({gf/tf*1e3:5.0f} TF)   dgrad {tb:7.1f} us ({gf/tb*1e3:5.0f} TF)", flush=True)
