"""How far the ORACLE train step itself amplifies tiny weight perturbations over 4 steps (justifies the multi-step tolerances in tests/test_model_gpu.py).  CPU only: python scripts/oracle_sensitivity.py"""
import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.torch_oracle import CycleGANOracle
def run(pert, pool):
    torch.manual_seed(17)
    o = CycleGANOracle(n_blocks=6, pool_size=pool, pool_seed=9)
    batches = [(torch.rand(2, 3, 64, 64) * 2 - 1, torch.rand(2, 3, 64, 64) * 2 - 1) for _ in range(4)]
    if pert:
        with torch.no_grad():
            for n in o.nets():
                for p in n.parameters():
                    p.mul_(1 + pert * torch.randn_like(p))
    out = []
    for step, (a, b) in enumerate(batches):
        if step == 2: o.set_epoch(150, 100, 100)
        out.append(o.train_step(a, b))
    return out
for pool in (0, 3):
    base = run(0, pool)
    for pert in (1e-6, 1e-5):
        r = run(pert, pool)
        for s in range(4):
            worst = max(abs(base[s][k] - r[s][k]) / max(1, abs(base[s][k])) for k in base[s])
            print(f"pool {pool} pert {pert:g} step {s}: worst rel diff {worst:.2e}")
