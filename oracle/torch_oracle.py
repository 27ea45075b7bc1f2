"""CPU oracle: the CycleGAN generator / discriminator / train step from stock torch.nn.

TEST INFRASTRUCTURE — not product code.  Nothing under unpaired-image-generation_amd/
imports this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Provenance.  The reference snapshot holds one file (/root/reference/README.md:1, a
title) and no source, tests or golden vectors, so there is no reference function to
follow line by line.  PARITY UNPINNED BY THE REFERENCE: the spec of record is
SURVEY.md Appendix A (architecture) + §3.1 (step order), restated here with the stock
torch CPU operators (torch==2.10.0+rocm7.0, ATen native + oneDNN) that define the
arithmetic at the operator boundary of SURVEY.md §8(b).  What pins this oracle instead:
  * SURVEY.md Appendix B known answers (param counts, config-1 output values / hashes,
    the 8 first-step losses)  -> tests/test_oracle.py
  * an independent fp64 numpy restatement of every operator (oracle/naive_fp64.py)
    -> tests/test_oracle.py
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResBlock(nn.Module):
    """x + [ReflPad1, Conv3x3, IN, ReLU, ReflPad1, Conv3x3, IN](x)   (SURVEY.md Appendix A)."""

    def __init__(self, dim: int):
        super().__init__()
        self.b = nn.Sequential(
            nn.ReflectionPad2d(1),
            nn.Conv2d(dim, dim, 3, 1, 0, bias=True),
            nn.InstanceNorm2d(dim, eps=1e-5, affine=False, track_running_stats=False),
            nn.ReLU(True),
            nn.ReflectionPad2d(1),
            nn.Conv2d(dim, dim, 3, 1, 0, bias=True),
            nn.InstanceNorm2d(dim, eps=1e-5, affine=False, track_running_stats=False),
        )

    def forward(self, x):
        return x + self.b(x)


def Generator(in_ch: int = 3, out_ch: int = 3, ngf: int = 64, n_blocks: int = 9) -> nn.Sequential:
    """ResNet generator, SURVEY.md Appendix A.  state_dict keys '{idx}.weight', '{idx}.b.{j}.weight'."""
    IN = lambda c: nn.InstanceNorm2d(c, eps=1e-5, affine=False, track_running_stats=False)
    layers = [nn.ReflectionPad2d(3), nn.Conv2d(in_ch, ngf, 7, 1, 0, bias=True), IN(ngf), nn.ReLU(True),
              nn.Conv2d(ngf, ngf * 2, 3, 2, 1, bias=True), IN(ngf * 2), nn.ReLU(True),
              nn.Conv2d(ngf * 2, ngf * 4, 3, 2, 1, bias=True), IN(ngf * 4), nn.ReLU(True)]
    layers += [ResBlock(ngf * 4) for _ in range(n_blocks)]
    layers += [nn.ConvTranspose2d(ngf * 4, ngf * 2, 3, 2, 1, output_padding=1, bias=True), IN(ngf * 2), nn.ReLU(True),
               nn.ConvTranspose2d(ngf * 2, ngf, 3, 2, 1, output_padding=1, bias=True), IN(ngf), nn.ReLU(True),
               nn.ReflectionPad2d(3), nn.Conv2d(ngf, out_ch, 7, 1, 0, bias=True), nn.Tanh()]
    return nn.Sequential(*layers)


def Discriminator(in_ch: int = 3, ndf: int = 64, n_layers: int = 3) -> nn.Sequential:
    """70x70 PatchGAN, SURVEY.md Appendix A (no sigmoid: LSGAN)."""
    IN = lambda c: nn.InstanceNorm2d(c, eps=1e-5, affine=False, track_running_stats=False)
    layers = [nn.Conv2d(in_ch, ndf, 4, 2, 1), nn.LeakyReLU(0.2, True)]
    nf = 1
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        layers += [nn.Conv2d(ndf * nf_prev, ndf * nf, 4, 2, 1, bias=True), IN(ndf * nf), nn.LeakyReLU(0.2, True)]
    nf_prev, nf = nf, min(2 ** n_layers, 8)
    layers += [nn.Conv2d(ndf * nf_prev, ndf * nf, 4, 1, 1, bias=True), IN(ndf * nf), nn.LeakyReLU(0.2, True)]
    layers += [nn.Conv2d(ndf * nf, 1, 4, 1, 1)]
    return nn.Sequential(*layers)


def init_weights(net: nn.Module) -> nn.Module:
    """conv / convT weights N(0, 0.02), biases 0, in net.modules() order (Appendix B recipe B1)."""
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.normal_(m.weight, 0.0, 0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return net


class OraclePool:
    """Image history buffer of the CycleGAN recipe [PAPER; Shrivastava et al.], SURVEY.md §8(f) row 1, restated with the
    stdlib RNG calls of the public recipe: while the pool fills every image is stored and returned; afterwards
    p = uniform(0, 1); p > 0.5: idx = randint(0, size - 1), the stored image is returned and replaced by the new one;
    otherwise the new image is returned.  One private random.Random(seed) per pool (the product seeds its pools the same
    way: pool_B with pool_seed, pool_A with pool_seed + 1).  size 0 = identity."""

    def __init__(self, size: int, seed: int):
        import random
        self.size, self.rng, self.images = int(size), random.Random(seed), []

    def query(self, images: torch.Tensor) -> torch.Tensor:
        if self.size == 0:
            return images
        out = []
        for img in images:
            img = img.detach().clone()
            if len(self.images) < self.size:
                self.images.append(img)
                out.append(img)
            elif self.rng.uniform(0, 1) > 0.5:
                idx = self.rng.randint(0, self.size - 1)
                out.append(self.images[idx])
                self.images[idx] = img
            else:
                out.append(img)
        return torch.stack(out)


class CycleGANOracle:
    """The §3.1 train step on stock torch CPU ops.  lambda=10, identity 0.5, LSGAN, Adam(2e-4, .5, .999).
    §8(f) rows 1-2: optional image pools in front of the discriminators' fake batch and the recipe's LR schedule as a stock
    torch LambdaLR on both optimisers (constant for n_const epochs, then linear to zero over n_decay)."""

    def __init__(self, n_blocks: int = 9, lr: float = 2e-4, lambda_cyc: float = 10.0, lambda_idt: float = 0.5,
                 dtype=torch.float32, pool_size: int = 0, pool_seed: int = 0):
        # Appendix B recipe B2: construct all four (default inits consume RNG), then re-init in this order.
        self.G_A, self.G_B = Generator(n_blocks=n_blocks), Generator(n_blocks=n_blocks)
        self.D_A, self.D_B = Discriminator(), Discriminator()
        for n in self.nets():
            init_weights(n).to(dtype)
        self.lam, self.lam_idt = lambda_cyc, lambda_idt
        self.opt_G = torch.optim.Adam(list(self.G_A.parameters()) + list(self.G_B.parameters()), lr=lr, betas=(0.5, 0.999))
        self.opt_D = torch.optim.Adam(list(self.D_A.parameters()) + list(self.D_B.parameters()), lr=lr, betas=(0.5, 0.999))
        self.pool_B, self.pool_A = OraclePool(pool_size, pool_seed), OraclePool(pool_size, pool_seed + 1)
        self._sched = None

    def set_epoch(self, epoch: int, n_const: int = 100, n_decay: int = 100):
        """LR of the 1-based training epoch `epoch` under the recipe's schedule: LambdaLR with
        lambda(e) = 1 - max(0, e + 1 - n_const) / (n_decay + 1), e = scheduler steps taken (the recipe's epoch_count = 1)."""
        import warnings
        from torch.optim.lr_scheduler import LambdaLR
        lam = lambda e: 1.0 - max(0, e + 1 - n_const) / float(n_decay + 1)
        if self._sched is None or self._sched[0] != (n_const, n_decay):
            self._sched = ((n_const, n_decay), [LambdaLR(o, lr_lambda=lam) for o in (self.opt_G, self.opt_D)])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # "scheduler.step() before optimizer.step()": the order is the caller's business here
            for sch in self._sched[1]:
                if sch.last_epoch > epoch - 1:
                    raise ValueError("the oracle's schedule only moves forward")
                while sch.last_epoch < epoch - 1:
                    sch.step()

    def nets(self):
        return (self.G_A, self.G_B, self.D_A, self.D_B)

    @staticmethod
    def _req(nets, flag):
        for n in nets:
            for p in n.parameters():
                p.requires_grad_(flag)

    def _run(self, net, x):
        """evaluate one of the four networks (hook: oracle/lowprec_oracle.py re-states the product's low-precision storage points here)"""
        return net(x)

    def train_step(self, real_A: torch.Tensor, real_B: torch.Tensor) -> dict:
        mse = lambda p, t: F.mse_loss(p, torch.full_like(p, t))
        run = self._run
        fake_B = run(self.G_A, real_A); rec_A = run(self.G_B, fake_B)
        fake_A = run(self.G_B, real_B); rec_B = run(self.G_A, fake_A)
        # --- generators (D frozen)
        self._req((self.D_A, self.D_B), False)
        self.opt_G.zero_grad()
        idt_A = run(self.G_A, real_B); idt_B = run(self.G_B, real_A)
        l_idt_A = F.l1_loss(idt_A, real_B) * self.lam * self.lam_idt
        l_idt_B = F.l1_loss(idt_B, real_A) * self.lam * self.lam_idt
        l_G_A = mse(run(self.D_A, fake_B), 1.0)
        l_G_B = mse(run(self.D_B, fake_A), 1.0)
        l_cyc_A = F.l1_loss(rec_A, real_A) * self.lam
        l_cyc_B = F.l1_loss(rec_B, real_B) * self.lam
        (l_G_A + l_G_B + l_cyc_A + l_cyc_B + l_idt_A + l_idt_B).backward()
        self.opt_G.step()
        # --- discriminators
        self._req((self.D_A, self.D_B), True)
        self.opt_D.zero_grad()
        pf_B, pf_A = self.pool_B.query(fake_B.detach()), self.pool_A.query(fake_A.detach())
        l_D_A = 0.5 * (mse(run(self.D_A, real_B), 1.0) + mse(run(self.D_A, pf_B), 0.0)); l_D_A.backward()
        l_D_B = 0.5 * (mse(run(self.D_B, real_A), 1.0) + mse(run(self.D_B, pf_A), 0.0)); l_D_B.backward()
        self.opt_D.step()
        self.last = dict(fake_B=fake_B.detach(), fake_A=fake_A.detach(), rec_A=rec_A.detach(), rec_B=rec_B.detach())
        return {"idt_A": l_idt_A.item(), "idt_B": l_idt_B.item(), "G_A": l_G_A.item(), "G_B": l_G_B.item(),
                "cyc_A": l_cyc_A.item(), "cyc_B": l_cyc_B.item(), "D_A": l_D_A.item(), "D_B": l_D_B.item()}
