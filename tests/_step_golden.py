"""Reader of the committed full-size train-step fixtures (tests/golden/step_*.npz, written by tests/golden/make_step_golden.py from
the CPU oracles in the build container).  The GPU tests regenerate weights and inputs from the seed exactly as the generator script
did and prove it with the stored checksums; then they compare the device step with the stored oracle results - the CPU oracle STEPS
(63 + 90 + 214 s on the GPU box's host cores) no longer run inside the -m gpu suite."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NS = 65536
LOSS_NAMES = ("idt_A", "idt_B", "G_A", "G_B", "cyc_A", "cyc_B", "D_A", "D_B")


def load(name):
    return np.load(os.path.join(GOLD, name))


def sample(t: torch.Tensor) -> torch.Tensor:
    """the generator script's sampling rule: every k-th element of the flattened tensor, at most NS of them"""
    f = t.detach().reshape(-1)
    k = max(1, -(-f.numel() // NS))
    return f[::k][:NS].float().cpu()


def assert_same_problem(gold, o, rA, rB):
    """weights (regenerated from the seed by constructing the oracle) and inputs are the ones the fixture was computed on"""
    w = torch.cat([p.detach().reshape(-1)[:64] for n in (o.G_A, o.G_B, o.D_A, o.D_B) for p in list(n.parameters())[:4]])
    mine = np.array([float(rA.double().sum()), float(rB.double().sum()), float(rA[0, 0, 0, 0]), float(w.double().sum()), float(w.abs().double().sum())])
    assert np.allclose(mine, gold["check"], rtol=1e-9, atol=1e-9), ("the seed no longer reproduces the fixture's weights / inputs", mine, gold["check"])


def losses(gold, key):
    return dict(zip(LOSS_NAMES, [float(v) for v in gold[key]]))


def rel_cos(mine: torch.Tensor, theirs: np.ndarray):
    """(relative L2, cosine) of a device tensor against a stored sample, on the sample's elements"""
    a, b = sample(mine), torch.from_numpy(theirs)
    return float((a - b).norm() / b.norm()), float(torch.nn.functional.cosine_similarity(a, b, dim=0))
