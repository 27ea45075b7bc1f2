"""Generator-only inference latency (Translator, one HIP graph per shape), with the inference InstanceNorm (statistics finalised
inside the apply kernel) against the training kernels (finalize launch + apply launch), alternated in one process.
python scripts/bench_infer.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
from unpaired_image_generation_amd.inference import Translator

def ev_time(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

g = u.Generator(n_blocks=9, dtype=torch.bfloat16)
res = {}
for rnd in range(3):
    for fused in (True, False):
        ops.INFER_FUSED_IN, ops.INFER_FUSED_MAX_BATCH = fused, 64
        for B, H, W in ((1, 256, 256), (2, 256, 256), (4, 256, 256), (1, 512, 512), (8, 256, 256)):
            x = torch.rand(B, H, W, 8, device="cuda").to(torch.bfloat16)
            tr = Translator(g, use_graph=True)
            ms = ev_time(lambda: tr.run_phys(x))
            res.setdefault((B, H, fused), []).append(ms)
            del tr
for (B, H, fused), v in sorted(res.items()):
    v = sorted(v)
    print(f"G9 bf16 B={B} {H}x{H} {'inference IN (finalize in apply)' if fused else 'training IN kernels          '}: median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f} ms")
