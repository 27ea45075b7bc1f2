"""Diagnostic: in-kernel stamps of wgrad_rows3_kernel (a library built with -DUIG_X_STAMP, loaded through UIG_LIB_PATH; the stamped
build overwrites six floats of every split's partial slab, so its results are NOT valid).  24 images, two networks:
per block of tile 0: [K loop cycles, K loop ns, epilogue cycles, epilogue ns, cycles waiting for DMA, cycles in the barrier].
UIG_LIB_PATH=ab/libuig_xstamp.so python scripts/stamp_wgrad_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
for B in (24, 16, 8):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.randn(B, 64, 64, 256, device="cuda") * 0.5).to(dt)
    for _ in range(20):
        pp = ops.conv_wgrad_pair_partial(l1.spec, x, dy, B // 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pp = ops.conv_wgrad_pair_partial(l1.spec, x, dy, B // 2)
    e1.record(); torch.cuda.synchronize()
    (ws1, splits), (ws2, _) = pp
    per = ws1.numel() // splits
    st = torch.stack([ws1[s * per: s * per + 6] for s in range(splits)] + [ws2[s * per: s * per + 6] for s in range(splits)]).cpu()
    steps = B // 2 * 64 / splits
    print(f"B={B}: {e0.elapsed_time(e1) * 1e3 / 20:6.1f} us per partial launch, splits {splits}, {steps:.1f} K-steps per block")
    names = ["K loop cycles", "K loop ns (x10 ns ticks)", "epilogue cycles", "epilogue ticks", "DMA-wait cycles", "barrier cycles"]
    for i, n in enumerate(names):
        print(f"   {n:28s} median {st[:, i].median():10.0f}  min {st[:, i].min():10.0f}  max {st[:, i].max():10.0f}")
    print(f"   cycles per K-step {float(st[:, 0].median()) / steps:7.0f}  (MFMA floor 1536)")
