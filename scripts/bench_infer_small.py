"""Generator-only inference latency (Translator, one HIP graph per shape): plain 3x3 launches of very small grids on the 64x64-tile
strip kernel (round 3; auto for <= 64 blocks of 128x128) against the 128x128-tile kernel, alternated in one process.
python scripts/bench_infer_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd.inference import Translator
lib = u.lib.lib()

def ev_time(fn, iters=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

g = u.Generator(n_blocks=9, dtype=torch.bfloat16)
res = {}
for rnd in range(3):
    for mode, name in ((2, "128x128 tiles (4 waves)       "), (0, "auto: 64x64 / 128x64 by grid   "), (3, "128x64 tiles wherever it applies")):
        u.ops.small_grid_kernels.MODE = mode         # what the Translator switches on around its launches
        for B, H, W in ((1, 256, 256), (2, 256, 256), (3, 256, 256), (4, 256, 256), (1, 512, 512)):
            x = torch.rand(B, H, W, 8, device="cuda").to(torch.bfloat16)
            tr = Translator(g, use_graph=True)
            res.setdefault((B, H, name), []).append(ev_time(lambda: tr.run_phys(x)))
            del tr
u.ops.small_grid_kernels.MODE = 0
for (B, H, name), v in sorted(res.items()):
    v = sorted(v)
    print(f"G9 bf16 B={B} {H}x{H} {name}: median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f} ms")
