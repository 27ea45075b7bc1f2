"""Diagnostic: in-kernel cycle stamps of the strip conv kernel (share of wait+barrier / DMA issue / reads+MFMA per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); layer.repack()
x = (torch.rand(8, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
import time
t0 = time.time()
while time.time() - t0 < 2.5:                      # hold the chip under load first: the clock it settles at is what matters
    for _ in range(200): ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
    torch.cuda.synchronize()
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

r0 = buf.view(256, 8, 8)[..., 1].double(); r1 = r0 + b[..., 7]
print("block entry skew: %.2f us (min..max of entry); first entry -> last exit %.2f us; median block life %.2f us" % (
    float(r0.max() - r0.min()) / 100, float(r1.max() - r0.min()) / 100, float(b[..., 7].median()) / 100))
def tm(n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
print("launch-to-launch, plain build   %.1f us" % tm())
u.lib.lib().uig_debug_set_strip_stamps(buf.data_ptr())
print("launch-to-launch, stamped build %.1f us" % tm())
u.lib.lib().uig_debug_set_strip_stamps(None)

# dead time between consecutive launches: last exit of launch k -> first entry of launch k+1 (absolute 100 MHz stamps)
bufs = [torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda") for _ in range(6)]
for _ in range(20): ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
for bb in bufs:
    u.lib.lib().uig_debug_set_strip_stamps(bb.data_ptr())
    ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias)
torch.cuda.synchronize()
u.lib.lib().uig_debug_set_strip_stamps(None)
ent = [bb.view(256, 8, 8)[..., 1].double() for bb in bufs]
ext = [e + bb.view(256, 8, 8)[..., 7].double() for e, bb in zip(ent, bufs)]
for k in range(1, 6):
    print("launch %d: first entry %.2f us after the previous launch's last exit; previous first-entry -> this first-entry %.2f us; grid life %.2f us" % (
        k, float(ent[k].min() - ext[k - 1].max()) / 100, float(ent[k].min() - ent[k - 1].min()) / 100, float(ext[k].max() - ent[k].min()) / 100))
bs = bufs[4].view(256, 8, 8).double()
print("steady state (5th back-to-back stamped launch): prologue %.0f | loop %.0f (wait %.0f) | epilogue %.0f | whole %.0f cycles; per-wave life %.2f us" % (
    bs[..., 4].median(), bs[..., 3].median(), bs[..., 0].median(), bs[..., 6].median(), bs[..., 5].median(), bs[..., 7].median() / 100))
