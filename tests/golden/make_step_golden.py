"""Fixtures of the three full-size train-step parity tests (BASELINE.json configs[1], [3], [4]) from the CPU oracles.

Run from the repo root (CPU only, ~15-25 minutes on 8 cores):  python tests/golden/make_step_golden.py [config2] [config4] [config5]

Why fixtures: the -m gpu tests used to run these CPU oracle steps on the GPU box (63 + 90 + 214 s of a 480-s suite against a 900-s
limit).  The oracle steps are deterministic functions of a seed (weights and inputs are regenerated from it: stock-torch CPU RNG and
ops, bitwise reproducible across thread counts), so their results are DATA: the 8 losses, strided 64 K-element
samples of the generated image and of three weight gradients, plus checksums of the regenerated inputs and weights so that a test can
prove it is looking at the same problem.  Written by the oracle in this container; no reference source exists to generate from
(/root/reference/README.md:1 is the whole tree): PARITY UNPINNED BY THE REFERENCE.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.lowprec_oracle import LowPrecOracle          # noqa: E402
from oracle.torch_oracle import CycleGANOracle           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NS = 65536
GRADS = (("G_A.14.b.5", lambda o: o.G_A[14].b[5].weight.grad), ("G_B.19", lambda o: o.G_B[19].weight.grad), ("D_A.8", lambda o: o.D_A[8].weight.grad))


def sample(t: torch.Tensor) -> np.ndarray:
    """every k-th element of the flattened tensor, at most NS of them (k = ceil(numel / NS)): the same rule in tests/_step_golden.py"""
    f = t.detach().reshape(-1)
    k = max(1, -(-f.numel() // NS))
    return f[::k][:NS].float().numpy().copy()


def checksum(o, rA, rB) -> np.ndarray:
    """what a test recomputes after regenerating weights and inputs from the seed"""
    w = torch.cat([p.detach().reshape(-1)[:64] for n in (o.G_A, o.G_B, o.D_A, o.D_B) for p in list(n.parameters())[:4]])
    return np.array([float(rA.double().sum()), float(rB.double().sum()), float(rA[0, 0, 0, 0]), float(w.double().sum()), float(w.abs().double().sum())])


LOSS_NAMES = ("idt_A", "idt_B", "G_A", "G_B", "cyc_A", "cyc_B", "D_A", "D_B")


def losses_arr(d) -> np.ndarray:
    return np.array([d[k] for k in LOSS_NAMES], dtype=np.float64)


def config2():
    """tests/test_model_gpu.py::test_train_step_config2_b4_256_bf16_graph_vs_oracle: seed 3, fp32 oracle steps 0 and 1, bf16 emulation step 0"""
    torch.manual_seed(3)
    o = CycleGANOracle(n_blocks=9)
    rA, rB = torch.rand(4, 3, 256, 256) * 2 - 1, torch.rand(4, 3, 256, 256) * 2 - 1
    torch.manual_seed(3)
    e = LowPrecOracle(n_blocks=9)
    out = dict(check=checksum(o, rA, rB))
    out["loss_fp32_step0"] = losses_arr(o.train_step(rA, rB))
    out["fake_B_fp32"] = sample(o.last["fake_B"])
    out["loss_fp32_step1"] = losses_arr(o.train_step(rA, rB))
    out["loss_emu_step0"] = losses_arr(e.train_step(rA, rB))
    out["fake_B_emu"] = sample(e.last["fake_B"])
    for name, get in GRADS:
        out["grad_emu_" + name] = sample(get(e))
    np.savez_compressed(os.path.join(OUT, "step_config2_b4_256_bf16.npz"), **out)


def config4():
    """tests/test_model_gpu.py::test_train_step_config4_b2_512_bf16_graph_vs_same_rounding_oracle: seed 13, bf16 emulation step 0"""
    torch.manual_seed(13)
    e = LowPrecOracle(n_blocks=9)
    rA, rB = torch.rand(2, 3, 512, 512) * 2 - 1, torch.rand(2, 3, 512, 512) * 2 - 1
    out = dict(check=checksum(e, rA, rB))
    out["loss_emu_step0"] = losses_arr(e.train_step(rA, rB))
    out["fake_B_emu"] = sample(e.last["fake_B"])
    out["grad_emu_G_A.14.b.5"] = sample(e.G_A[14].b[5].weight.grad)
    np.savez_compressed(os.path.join(OUT, "step_config4_b2_512_bf16.npz"), **out)


def config5():
    """tests/test_fp8_gpu.py::test_train_step_config5_b8_256_fp8_graph_vs_same_rounding_oracle: seed 4, MX-fp8 emulation step 0"""
    torch.manual_seed(4)
    o = LowPrecOracle(n_blocks=9, fp8=True)
    rA, rB = torch.rand(8, 3, 256, 256) * 2 - 1, torch.rand(8, 3, 256, 256) * 2 - 1
    out = dict(check=checksum(o, rA, rB))
    out["loss_emu_step0"] = losses_arr(o.train_step(rA, rB))
    out["fake_B_emu"] = sample(o.last["fake_B"])
    for name, get in GRADS:
        out["grad_emu_" + name] = sample(get(o))
    np.savez_compressed(os.path.join(OUT, "step_config5_b8_256_fp8.npz"), **out)


if __name__ == "__main__":
    import time
    todo = sys.argv[1:] or ["config2", "config4", "config5"]
    for name in todo:
        t0 = time.time()
        globals()[name]()
        print(f"{name}: {time.time() - t0:.0f} s", flush=True)
