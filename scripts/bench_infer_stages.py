"""Generator-only inference latency (Translator, one HIP graph per shape): the 128x128-tile strip kernel with two weight stages against
four (tiles fetched three K-steps ahead; opt-in, measured slower), alternated in one process; plus a serialised
per-kernel-family split of one batch-1 call from HIP events is left to rocprofv3 (scripts/r3_prof_infer.sh).
python scripts/bench_infer_stages.py"""
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
    for stages in (2, 4):
        lib.uig_debug_set_strip_stages(stages)
        for B, H, W in ((1, 256, 256), (2, 256, 256), (1, 512, 512)):
            x = torch.rand(B, H, W, 8, device="cuda").to(torch.bfloat16)
            tr = Translator(g, use_graph=True)
            res.setdefault((B, H, stages), []).append(ev_time(lambda: tr.run_phys(x)))
            del tr
lib.uig_debug_set_strip_stages(2)
for (B, H, stages), v in sorted(res.items()):
    v = sorted(v)
    print(f"G9 bf16 B={B} {H}x{H} weight stages {stages}: median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f} ms")
