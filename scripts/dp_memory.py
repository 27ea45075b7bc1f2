"""Peak device memory of the train step (batch 4, 256x256, bf16) with the backward passes staged (data-parallel option) against the
un-staged step, eager and graph replay: torch.cuda.max_memory_allocated after 3 steps (staged_backward keeps every
stage's saved tensors until the step ends).  python scripts/dp_memory.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u

rA, rB = (torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1 for _ in range(2))
for graph in (False, True):
    for staged in (False, True):
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=graph, stage_backward=staged)
        for _ in range(3):
            m.train_step(rA, rB)
        torch.cuda.synchronize()
        print(f"{'graph' if graph else 'eager'} {'staged 4+2' if staged else 'un-staged '}: peak allocated {(torch.cuda.max_memory_allocated() - base) / 2**30:6.2f} GiB, "
              f"reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB", flush=True)
        m.close(); del m
