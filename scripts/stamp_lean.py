"""Diagnostic: shader cycles of the phased strip schedule (the default; stamped build: entry / K-loop start / end / epilogue stamps only)
against its wall time: cycles per K-step and the clock the launch actually runs at.  python scripts/stamp_lean.py [B=16] [dm=0]   (dm 5: round 3's schedule)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
DM = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.uig_debug_set_strip_pk(DM, 0)
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
STATS = "nostats" not in sys.argv      # `nostats`: the same launch without the fused InstanceNorm statistics (what they cost the epilogue)
f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=STATS)
def t(n=200):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
us = t()
NB = 256
buf = torch.zeros(NB * 8 * 8 + NB * 8 * 10, dtype=torch.int64, device="cuda")
for _ in range(200): f()
lib.uig_debug_set_strip_stamps(buf.data_ptr())
for _ in range(50): f()
torch.cuda.synchronize()
us_st = t(50)
lib.uig_debug_set_strip_stamps(None)
st = buf[:NB * 8 * 8].view(NB, 8, 8).double()
ok = st[..., 0].min(dim=1).values > 0
st = st[ok]
# stamps: 0 entry, 1 K start, 2 K end, 3 (epilogue) ..., 6 K start of tile 1, 7 K end of tile 1
k0, k1 = st[..., 2] - st[..., 1], st[..., 7] - st[..., 6]
tot = st[..., 7].amax(dim=1) - st[..., 0].amin(dim=1)      # entry -> end of the second K loop, per block
print(f"DM {DM} B={B}: product build {us:.1f} us, stamped build {us_st:.1f} us per launch")
print(f"  K loop tile 0: {float(k0.median()):.0f} cycles = {float(k0.median()) / 36:.0f} per step; tile 1: {float(k1.median()):.0f} = {float(k1.median()) / 36:.0f} per step")
print(f"  epilogue of tile 0: accumulators -> LDS {float((st[..., 4] - st[..., 3]).median()):.0f}, rows -> statistics + stores {float((st[..., 5] - st[..., 4]).median()):.0f} cycles (stamps 3 / 4 / 5)")
sm = buf[NB * 8 * 8:NB * 8 * 12].view(NB, 8, 4).double()[ok]
print(f"  prologue in parts (cycles from entry): first DMAs issued {float(sm[..., 0].median()):.0f}, row table built {float(sm[..., 1].median()):.0f}, accumulators initialised = K start {float((st[..., 1] - st[..., 0]).median()):.0f}, first DMAs landed {float(sm[..., 2].median()):.0f}")
print(f"  prologue (entry -> K start): {float((st[..., 1] - st[..., 0]).median()):.0f}; between the K loops: {float((st[..., 6] - st[..., 2]).median()):.0f}; entry -> end of K loop 1: {float(tot.median()):.0f} cycles")
print(f"  => if the stamped launch is ~{us_st:.1f} us and ~{float(tot.median()) + 7500:.0f} cycles long: clock ~{(float(tot.median()) + 7500) / us_st / 1e3:.2f} GHz")
lib.uig_debug_set_strip_pk(0, 0)
