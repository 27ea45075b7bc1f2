"""Diagnostic: timing ablations of the phased strip schedule (conv_strip_pk.hip, DM 6): the same launch with parts of the K-step left out
(wrong results - timing only), in microseconds (product build) and in shader cycles per K-step (stamped build: K-loop start / end stamps
only, so the clock the variant happens to run at drops out).  python scripts/ablate_phased.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
B = 16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
def t(n=60):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
NB = 256
buf = torch.zeros(NB * 8 * 8 + NB * 8 * 10, dtype=torch.int64, device="cuda")
def cycles():
    for _ in range(20): f()
    buf.zero_()
    lib.uig_debug_set_strip_stamps(buf.data_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize()
    lib.uig_debug_set_strip_stamps(None)
    st = buf[:NB * 8 * 8].view(NB, 8, 8).double()
    ok = st[..., 0].min(dim=1).values > 0
    st = st[ok]
    k = (st[..., 2] - st[..., 1]) + (st[..., 7] - st[..., 6])       # stamps: 1 / 2 = K loop of tile 0, 6 / 7 = of tile 1
    return float(k.median()) / 72, float((st[..., 7].max() - st[..., 0].min()))
CASES = [("phased", 6), ("phased, no DMAs", 6 | 1 << 8), ("phased, vmcnt(0)", 6 | 2 << 8), ("phased, no strip reads", 6 | 4 << 8),
         ("phased, no weight reads", 6 | 8 << 8), ("phased, no fragment reads", 6 | 12 << 8), ("phased, no reads, no DMAs", 6 | 13 << 8), ("phased, no MFMAs", 6 | 16 << 8),
         ("phased, no priority", 6 | 32 << 8), ("phased, no MFMAs, no reads", 6 | 28 << 8), ("phased, nothing but barriers", 6 | 29 << 8)]
res, cyc = {}, {}
for rnd in range(3):
    for name, dm in CASES:
        lib.uig_debug_set_strip_pk(dm, 0)
        res.setdefault(name, []).append(t())
        cyc.setdefault(name, []).append(cycles())
lib.uig_debug_set_strip_pk(0, 0)
for name, _ in CASES:
    v = sorted(res[name]); c = sorted(cyc[name])
    print(f"{name:34s} {v[1]:6.1f} us (min {v[0]:6.1f})   stamped build: {c[1][0]:6.0f} cycles per K-step, launch {c[1][1]:8.0f} cycles")
