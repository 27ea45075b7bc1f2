"""Worker of tests/test_model_gpu.py::test_process_group_lifecycle_then_new_graph_model: ONE process that creates a process
group, trains under it (HIP graphs + forced RCCL exchange), tears the model and the group down and then trains a NEW
graph-captured model - the sequence a notebook, a sweep or an elastic restart performs, and the one that crashed inside
CUDAGraph.replay in round 2 (gpurun_out/r2d_tests.log).  faulthandler is on so that a native fault leaves the Python
stack of every thread; exits 0 through the normal interpreter shutdown after printing PG_LIFECYCLE_OK."""
import faulthandler
import os
import socket
import sys

faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import unpaired_image_generation_amd as u


def steps(m, ref, rA, rB, n):
    for step in range(n):
        a, b = ref.train_step(rA, rB), m.train_step(rA, rB)
        assert a == b, (step, a, b)


def main():
    torch.cuda.set_device(0)
    torch.manual_seed(21)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    for cycle in range(2):                                           # two full group lifetimes in one process
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        torch.manual_seed(5)
        m0 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=True)
        m1 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=True, force_exchange=True)
        m1.load_state_dicts(*[n.state_dict() for n in m0.nets()])
        steps(m1, m0, rA, rB, 2)
        assert m1.graph_active and m1.xchg.n_started >= 4
        m1.close(); m0.close()
        assert u.ops.device_state_empty(), "state of a closed model survives in the operator layer"
        dist.barrier()
        dist.destroy_process_group()
        print(f"PG_CYCLE_{cycle}_DOWN", flush=True)
        # a new model, captured and replayed with no group alive: eager and graph must agree bitwise
        torch.manual_seed(6)
        me = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=False)
        mg = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=True)
        mg.load_state_dicts(*[n.state_dict() for n in me.nets()])
        steps(mg, me, rA, rB, 2)
        assert mg.graph_active and mg.xchg.world == 1 and not mg.xchg.active
        mg.close(); me.close()
        print(f"PG_CYCLE_{cycle}_RETRAINED", flush=True)
    print("PG_LIFECYCLE_OK", flush=True)


if __name__ == "__main__":
    main()
