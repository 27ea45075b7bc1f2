"""CPU emulation of the MX block-scaled fp8 convolution path (BASELINE.json configs[4]): same rounding as the device.

TEST INFRASTRUCTURE - not product code (only tests/ import this).  No reference implementation exists
(/root/reference/README.md:1 is the whole tree): PARITY UNPINNED BY THE REFERENCE.  What is restated here is the OCP
Microscaling (MX) format as the product uses it - e4m3 elements (OCP "e4m3fn": bias 7, max 448, no infinities), one E8M0
power-of-two scale per 32 consecutive channels, scale = 2^(floor(log2(max|x|)) - 8) - with stock torch's
float8_e4m3fn cast (round to nearest even) as the element rounding.  The device quantiser (uig_mx_quantize) must match
`mx_quantize` byte for byte; the device convolution (uig_conv3x3_mx_fp8, fp32 accumulate) must match `F.conv2d` on the
de-quantised operands up to fp32 summation order and the bf16 rounding of its output.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BLOCK = 32
EMAX_E4M3 = 8            # 448 = 1.75 * 2^8


def mx_quantize(x: torch.Tensor):
    """x (..., C), C % 32 == 0, float32 or bfloat16 -> (q uint8 (..., C) e4m3 bytes, s uint8 (..., C/32) E8M0 bytes)."""
    xf = x.detach().float().contiguous()
    C = xf.shape[-1]
    assert C % BLOCK == 0
    blk = xf.reshape(*xf.shape[:-1], C // BLOCK, BLOCK)
    amax = blk.abs().amax(-1)
    eb = (amax.view(torch.int32) >> 23) & 0xFF                        # biased exponent = floor(log2(amax)) + 127 (0 for 0 / subnormals)
    sb = torch.where(amax == 0, torch.full_like(eb, 127), (eb - EMAX_E4M3).clamp(0, 254))
    inv = torch.ldexp(torch.ones_like(amax), 127 - sb)                # 2^-(sb - 127), exact
    v = (blk * inv.unsqueeze(-1)).clamp(-448.0, 448.0)
    q = v.to(torch.float8_e4m3fn).view(torch.uint8).reshape(xf.shape)
    return q, sb.to(torch.uint8)


def mx_dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    C = q.shape[-1]
    v = q.view(torch.float8_e4m3fn).float().reshape(*q.shape[:-1], C // BLOCK, BLOCK)
    scale = torch.ldexp(torch.ones(s.shape, dtype=torch.float32), s.to(torch.int32) - 127)
    return (v * scale.unsqueeze(-1)).reshape(q.shape)


def fake_quant_channels(x_nchw: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> the values the device convolution sees: MX-quantised along C per pixel, de-quantised, fp32."""
    xl = x_nchw.permute(0, 2, 3, 1).contiguous()
    return mx_dequantize(*mx_quantize(xl)).permute(0, 3, 1, 2).contiguous()


def fake_quant_weight(w: torch.Tensor, axis: int) -> torch.Tensor:
    """Conv weight (Cout, Cin, kH, kW) quantised along `axis` (1: forward operand [Cout][tap][Cin]; 0: input-gradient
    operand [Cin][tap][Cout]) in blocks of 32, de-quantised.  The device quantises the bf16 packed operand, so the weight
    is rounded to bf16 first."""
    wb = w.detach().to(torch.bfloat16).float()
    wl = wb.movedim(axis, -1).contiguous()
    return mx_dequantize(*mx_quantize(wl)).movedim(-1, axis).contiguous()


def conv3x3_mx_forward(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, reflect: bool) -> torch.Tensor:
    """3x3 stride-1 pad-1 conv as the fp8 path computes it (fp32 result, before the bf16 output rounding)."""
    xq = fake_quant_channels(x)
    xq = F.pad(xq, (1, 1, 1, 1), mode="reflect") if reflect else F.pad(xq, (1, 1, 1, 1))
    return F.conv2d(xq, fake_quant_weight(w, 1), b)


def conv3x3_mx_dgrad_zero_pad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """input gradient of a ZERO-padded 3x3 stride-1 conv from MX-quantised dy and the dgrad weight operand (the main term of
    the reflection-pad gradient: the mirrored-border terms stay on the bf16 path)"""
    return F.conv_transpose2d(fake_quant_channels(dy), fake_quant_weight(w, 0), None, 1, 1)


def conv3x3_mx_dgrad_reflect_mirror(dy: torch.Tensor, w: torch.Tensor, quant: bool = True) -> torch.Tensor:
    """input gradient of a pad-1 REFLECTION 3x3 stride-1 conv as the one-launch fp8 kernel (uig_conv3x3_mx_fp8_dgrad_mirror) computes it:
    a zero-padded transposed conv of the MX-quantised dy in which the taps that would touch a mirrored line / column read a "mirror
    pixel" = the fp32 sum of two (four at the corners) de-quantised pixels, RE-QUANTISED per 32-channel block.  quant=False: no
    quantisation anywhere - then the result must equal autograd's gradient of F.pad(reflect)+conv2d (the algebra check in tests/)."""
    fq = fake_quant_channels if quant else (lambda t: t)
    q = fq(dy.float())
    wq = fake_quant_weight(w, 0) if quant else w.float()
    B, C, H, W = q.shape

    def sh(X, dh, dw):                       # Y[h][w] = X[h + dh][w + dw], zero outside
        hh, ww = X.shape[2:]
        P = F.pad(X, (1, 1, 1, 1))
        return P[:, :, 1 + dh:1 + dh + hh, 1 + dw:1 + dw + ww]

    px = lambda r, c: q[:, :, r:r + 1, c:c + 1]
    mcl, mcr = fq(q[..., 2:3] + q[..., 0:1]), fq(q[..., W - 3:W - 2] + q[..., W - 1:W])            # column mirrors, one per row
    mrt, mrb = fq(q[:, :, 2:3] + q[:, :, 0:1]), fq(q[:, :, H - 3:H - 2] + q[:, :, H - 1:H])        # line mirrors, one per column
    corner = lambda ra, rb, ca, cb: fq(((px(ra, ca) + px(rb, ca)) + px(ra, cb)) + px(rb, cb))      # the device's summation order
    dx = torch.zeros(B, w.shape[1], H, W)
    for kh in range(3):
        for kw in range(3):
            dh, dw = 1 - kh, 1 - kw
            S = sh(q, dh, dw).clone()
            if dw == 1:
                S[:, :, :, 1:2] = sh(mcl, dh, 0)
            if dw == -1:
                S[:, :, :, W - 2:W - 1] = sh(mcr, dh, 0)
            if dh != 0:
                row, m = (1, mrt) if dh == 1 else (H - 2, mrb)
                ra, rb = (2, 0) if dh == 1 else (H - 3, H - 1)
                S[:, :, row:row + 1, :] = sh(m, 0, dw)
                if dw == 1:
                    S[:, :, row:row + 1, 1:2] = corner(ra, rb, 2, 0)
                if dw == -1:
                    S[:, :, row:row + 1, W - 2:W - 1] = corner(ra, rb, W - 3, W - 1)
            dx = dx + torch.einsum("bohw,oi->bihw", S, wq[:, :, kh, kw])
    return dx
