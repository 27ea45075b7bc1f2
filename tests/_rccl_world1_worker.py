"""Worker of tests/test_model_gpu.py::test_graph_step_with_rccl_exchange_world1_and_close: one rank of a 1-rank RCCL job.
Exits 0 through the normal interpreter shutdown (no os._exit) after printing RCCL_WORLD1_OK."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import unpaired_image_generation_amd as u


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.manual_seed(21)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    torch.manual_seed(5)
    m0 = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=True)
    m1 = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=True, force_exchange=True)
    m2 = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=True, force_exchange=True, stage_backward=True)   # the staged form: a graph per backward stage
    for m in (m1, m2):
        m.load_state_dicts(*[n.state_dict() for n in m0.nets()])
    assert m1.xchg.force and m0.xchg.world == 1 and (len(m1.buckets_G), len(m2.buckets_G), len(m2.buckets_D)) == (1, 4, 2)
    for step in range(3):
        l0, l1, l2 = m0.train_step(rA, rB), m1.train_step(rA, rB), m2.train_step(rA, rB)
        assert m1.graph_active and m0.graph_active and m2.graph_active
        assert l0 == l1 == l2, (step, l0, l1, l2)
    for m in (m1, m2):
        assert torch.equal(m0.grp_G.flat, m.grp_G.flat) and torch.equal(m0.grp_D.flat, m.grp_D.flat)
    assert m1.xchg.n_started >= 6 and m2.xchg.n_started >= 18 and m0.xchg.n_started == 0      # the all-reduces really went through RCCL
    m2.close(); m1.close(); m0.close()
    assert m1._graphs is None and m2._graphs is None
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)


if __name__ == "__main__":
    main()
