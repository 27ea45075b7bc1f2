"""Diagnostic: in-kernel cycle stamps of the strip conv kernel (share of wait+barrier / DMA issue / reads+MFMA per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); layer.repack()
x = (torch.rand(8, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
for _ in range(3): ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
u.lib.lib().uig_debug_set_strip_stamps(buf.data_ptr())
for _ in range(3): ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
torch.cuda.synchronize()
u.lib.lib().uig_debug_set_strip_stamps(None)
b = buf.view(256, 8, 8).double()
tot = b[..., 3]
print("per-wave total cycles: median %.0f min %.0f max %.0f" % (tot.median(), tot.min(), tot.max()))
for i, n in enumerate(("wait+barrier", "dma issue", "reads+mfma")):
    print(f"{n:14s} median {b[..., i].median():9.0f} cycles = {100 * float((b[..., i] / tot).median()):5.1f} % of the loop; per K-step {b[..., i].median() / 36:7.0f}")

print("prologue cycles median %.0f | epilogue (store+drain) %.0f | whole kernel %.0f cycles = %.2f us (realtime) -> clock %.2f GHz" % (
    b[..., 4].median(), b[..., 6].median(), b[..., 5].median(), b[..., 7].median() / 100.0, float((b[..., 5] / (b[..., 7] / 100.0)).median()) / 1e3))
