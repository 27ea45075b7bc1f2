"""Cost of the in-launch statistics finalize (round 4) on the dominant launch: conv forward (+ final statistics) with the finalize
launch behind it (hook 0), with arrival tickets (1), and the diagnostic forms (2: tickets taken, reduction skipped; 3: write-through
slab stores + drain + barriers only).  Interleaved rounds in one process.  python scripts/bench_tickets.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
def t(fn, n=60):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
res = {}
for rnd in range(3):
    for B in (16, 8):
        x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
        for mode in (-1, 0, 1, 2, 3):
            if mode < 0:
                ops.IN_TICKETS = False
            else:
                ops.IN_TICKETS = True; lib.uig_debug_set_in_tickets(mode)
            with torch.no_grad():
                f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
                res.setdefault((B, mode), []).append(t(f))
    lib.uig_debug_set_in_tickets(1)
names = {-1: "partials only (no finalize at all)", 0: "conv + finalize LAUNCH", 1: "in-launch finalize (tickets)", 2: "tickets, reduction skipped", 3: "sc1 slabs + drain + barriers only"}
for (B, mode), xs in sorted(res.items()):
    print(f"B{B:2d} {names[mode]:40s} {sorted(xs)[len(xs)//2]:7.2f} us   {['%.1f' % v for v in xs]}")
