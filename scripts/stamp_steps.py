"""Diagnostic (STAMP build of the persistent strip kernel, default schedule): where a K-step's cycles go, per wave - sums over all K-steps
of the cycles spent in the step's `s_waitcnt` (own DMAs / fragment reads) and in its barrier, against the K loops' total.
python scripts/stamp_steps.py [B=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
lib.uig_debug_set_strip_pk(5, 0)      # round 3's schedule (the per-step sums exist in its stamped build only; the phased default: scripts/stamp_lean.py)
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(100): f()
    torch.cuda.synchronize()
NB = 256
buf = torch.zeros(NB * 8 * 8 + NB * 8 * 4, dtype=torch.int64, device="cuda")
lib.uig_debug_set_strip_stamps(buf.data_ptr())
for _ in range(3): f()
torch.cuda.synchronize()
lib.uig_debug_set_strip_stamps(None)
st = buf[:NB * 8 * 8].view(NB, 8, 8).double()
sm = buf[NB * 8 * 8:].view(NB, 8, 4).double()
ok = st[..., 0].min(dim=1).values > 0
st, sm = st[ok], sm[ok]
# stamps of the STAMP build: 0 entry, 1 K start (tile 0), 2 K end, 3 behind the barrier, 4 accumulators in LDS, 5 epilogue issued, 6 K start (tile 1), 7 K end
kloop = (st[..., 2] - st[..., 1]) + ((st[..., 7] - st[..., 6]) if B >= 16 else 0)
steps = sm[..., 2]
print(f"B={B}: blocks {int(ok.sum())}, K-steps per wave {steps.median():.0f}, K loops {kloop.median():.0f} cycles = {float((kloop / steps).median()):.0f} per step")
for name, w in (("issuing waves 0-3", slice(0, 4)), ("other waves 4-7", slice(4, 8))):
    ws, bs, n, kl = sm[:, w, 0], sm[:, w, 1], sm[:, w, 2], kloop[:, w]
    print(f"  {name}: waitcnt {float((ws / n).median()):6.0f} cycles/step, barrier {float((bs / n).median()):6.0f}, everything else {float(((kl - ws - bs) / n).median()):6.0f}")
