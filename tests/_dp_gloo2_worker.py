"""Worker of tests/test_model_gpu.py::test_two_rank_data_parallel_step_equals_full_batch: one of two ranks of a gloo job that share
GPU 0 (RCCL refuses two ranks on one device; gloo moves the CUDA buffers through the host, which is all this numerics test needs).
argv: rank port outfile graph(0/1) staged(0/1).  Runs 2 data-parallel train steps of the product's own step (1/world folded into
Adam; staged = 1: backward in stages, bucketed all-reduce between the stage graphs; staged = 0, the default of round 3: one all-reduce
per optimiser group behind its phase) on ITS half of a fixed batch and saves the summed gradient buffers of the first step and the
parameters after the second."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import unpaired_image_generation_amd as u


def main():
    rank, port, out, graph = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1"
    staged = len(sys.argv) > 5 and sys.argv[5] == "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    torch.manual_seed(31)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))      # the full batch; this rank trains on image `rank`
    torch.manual_seed(9)
    m = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph, stage_backward=True if staged else None)   # same seed on both ranks = identical replicas
    m.broadcast_params(0)
    assert m.xchg.world == 2 and m.xchg.active and (len(m.buckets_G), len(m.buckets_D)) == ((4, 2) if staged else (1, 1))
    a, b = rA[rank:rank + 1].contiguous(), rB[rank:rank + 1].contiguous()
    m.train_step(a, b)
    gG, gD = m.grp_G.grad.clone(), m.grp_D.grad.clone()                               # SUM over the ranks (1/world lives in Adam)
    m.train_step(a, b)
    torch.cuda.synchronize()
    assert m.xchg.n_started >= (12 if staged else 4)
    torch.save({"gG": gG.cpu(), "gD": gD.cpu(), "pG": m.grp_G.flat.cpu(), "pD": m.grp_D.flat.cpu()}, out)
    m.close()
    dist.barrier()
    dist.destroy_process_group()
    print("DP_GLOO2_OK", rank, flush=True)


if __name__ == "__main__":
    main()
