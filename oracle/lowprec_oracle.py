"""Same-rounding CPU emulation of the bf16 and the MX-fp8 mixed-precision train step (BASELINE.json configs[1] / configs[4]).

TEST INFRASTRUCTURE - not product code (only tests/ import this).  No reference implementation exists
(/root/reference/README.md:1 is the whole tree): PARITY UNPINNED BY THE REFERENCE.  This file restates the product's
low-precision STORAGE POINTS on top of the stock-torch oracle (oracle/torch_oracle.py, which defines the architecture, the
step order and the fp32 arithmetic) so that a low-precision step can be checked against something tighter than "within a
few percent of fp32":

  * every tensor the device keeps in HBM between two kernels is bf16: network inputs, every convolution output (bias and
    the epilogue activation - LeakyReLU of the first PatchGAN layer, tanh of the generator head - applied in fp32 before the
    one rounding), every InstanceNorm(+ReLU/LeakyReLU)(+residual add) output (one rounding after the add), and the
    gradient of each of those tensors in the backward pass (`_Store`: rounds the value forward and its gradient backward);
  * convolution operands are those bf16 tensors and the bf16-rounded weights (`_OperandW`: rounds forward, passes the
    gradient through untouched - weight gradients, Adam and the master weights are fp32), fp32 accumulate;
  * fp8=True: the generators' ResBlock convolutions run forward on MX block-scaled e4m3 operands and their INPUT gradient's
    main (zero-padded) term on the MX-quantised dy and the MX-quantised input-gradient weight operand, exactly as
    oracle/mx_fp8.py states them; the mirrored-border terms of that gradient and the weight / bias gradients stay on the
    bf16 operands (`_MXConv3x3`), as the product does.

What this emulation does NOT reproduce: the fp32 summation ORDER inside a kernel (MFMA K order, split-K slabs, wave
reductions) - so results agree to fp32 rounding of each accumulation followed by the same bf16 rounding, which occasionally
lands on the other side of a bf16 rounding boundary; tolerances in the tests are stated accordingly.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mx_fp8 as M
from .torch_oracle import CycleGANOracle, ResBlock


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _Store(torch.autograd.Function):
    """a tensor written to HBM as bf16: value rounded forward, gradient rounded backward"""

    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


class _OperandW(torch.autograd.Function):
    """the bf16 kernel-side copy of an fp32 master weight: rounded forward, fp32 gradient passed through"""

    @staticmethod
    def forward(ctx, w):
        return _bf(w)

    @staticmethod
    def backward(ctx, g):
        return g


store = _Store.apply
wop = _OperandW.apply


def mirror_dgrad_shape(shape) -> bool:
    """the product's dispatch rule (uig_conv3x3_mx_fp8_dgrad_mirror_applicable): which fp8 layers take the one-launch input gradient"""
    B, C, H, W = shape
    return W == 64 and H >= 8 and H % 4 == 0 and C % 128 == 0


class _MXConv3x3(torch.autograd.Function):
    """ReflectionPad2d(1) + Conv2d(k=3) of a ResBlock on the MX fp8 path.  x: bf16-representable activations, w fp32 master."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return M.conv3x3_mx_forward(x, w, b, True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        wb = _bf(w)
        if mirror_dgrad_shape(x.shape):
            # 64-wide maps (the 256x256 configs): one launch, the mirrored terms as re-quantised mirror pixels (uig_conv3x3_mx_fp8_dgrad_mirror)
            dx = M.conv3x3_mx_dgrad_reflect_mirror(dy, w)
        else:
            # input gradient: fp8 main term (zero-padded transposed conv of the quantised dy) + the exact mirrored-border terms on bf16 operands
            main = M.conv3x3_mx_dgrad_zero_pad(dy, w)
            with torch.enable_grad():
                xr = x.detach().requires_grad_(True)
                full = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wb)
                (gx_full,) = torch.autograd.grad(full, xr, dy)
            zp = F.conv_transpose2d(dy, wb, None, 1, 1)
            dx = main + (gx_full - zp)
        # weight / bias gradients: bf16 operands (the un-quantised x and dy), fp32 result
        dw = torch.nn.grad.conv2d_weight(F.pad(x, (1, 1, 1, 1), mode="reflect"), w.shape, dy)
        return dx, dw, dy.sum((0, 2, 3))


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def emu_generator(net: nn.Sequential, x: torch.Tensor, fp8: bool = False) -> torch.Tensor:
    """forward of an oracle Generator (torch_oracle.Generator) with the product's bf16 storage points; fp8: ResBlock convs on MX fp8"""
    mods = list(net)
    h = x                                        # the caller passes a stored (bf16-representable) tensor
    i = 0
    pad = None
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReflectionPad2d):
            pad = m.padding[0]
            i += 1
        elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            hp = F.pad(h, (pad,) * 4, mode="reflect") if pad else h
            pad = None
            if isinstance(m, nn.Conv2d):
                y = F.conv2d(hp, wop(m.weight), m.bias, m.stride, m.padding)
            else:
                y = F.conv_transpose2d(hp, wop(m.weight), m.bias, m.stride, m.padding, m.output_padding)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, nn.Tanh):         # epilogue activation: one rounding after it
                h = store(torch.tanh(y)); i += 2
            else:
                h = store(y); i += 1
        elif isinstance(m, nn.InstanceNorm2d):   # IN + ReLU fused: one rounding
            assert isinstance(mods[i + 1], nn.ReLU)
            h = store(F.relu(_inorm(h))); i += 2
        elif isinstance(m, ResBlock):
            c1, c2 = m.b[1], m.b[5]
            if fp8:
                a = store(_MXConv3x3.apply(h, c1.weight, c1.bias))
                a = store(F.relu(_inorm(a)))
                a = store(_MXConv3x3.apply(a, c2.weight, c2.bias))
            else:
                a = store(F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), wop(c1.weight), c1.bias))
                a = store(F.relu(_inorm(a)))
                a = store(F.conv2d(F.pad(a, (1, 1, 1, 1), mode="reflect"), wop(c2.weight), c2.bias))
            h = store(h + _inorm(a))             # IN + residual add fused: one rounding
            i += 1
        else:
            raise TypeError(f"emu_generator: unexpected module {type(m).__name__} at {i}")
    return h


def emu_discriminator(net: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    mods = list(net)
    h = x
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            y = F.conv2d(h, wop(m.weight), m.bias, m.stride, m.padding)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, nn.LeakyReLU):    # first layer: LeakyReLU in the conv epilogue
                h = store(F.leaky_relu(y, nxt.negative_slope)); i += 2
            else:
                h = store(y); i += 1
        elif isinstance(m, nn.InstanceNorm2d):
            act = mods[i + 1]
            assert isinstance(act, nn.LeakyReLU)
            h = store(F.leaky_relu(_inorm(h), act.negative_slope)); i += 2
        else:
            raise TypeError(f"emu_discriminator: unexpected module {type(m).__name__} at {i}")
    return h


class LowPrecOracle(CycleGANOracle):
    """CycleGANOracle whose networks are evaluated with the product's low-precision storage points (module docstring).
    Same constructor, same weights for the same seed, same train_step (the parent's, through its `_run` hook)."""

    def __init__(self, n_blocks: int = 9, fp8: bool = False, **kw):
        super().__init__(n_blocks=n_blocks, **kw)
        self.fp8 = bool(fp8)

    def _run(self, net, x):
        x = store(x)                                             # inputs reach the device as bf16 (to_nhwc)
        if net is self.G_A or net is self.G_B:
            return emu_generator(net, x, self.fp8)
        return emu_discriminator(net, x)

    def train_step(self, real_A, real_B):
        return super().train_step(_bf(real_A), _bf(real_B))
