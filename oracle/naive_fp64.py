"""Independent fp64 numpy restatement of every operator on the hot path, from the
operator definitions (SURVEY.md §8(b) ATen schema semantics).  TEST INFRASTRUCTURE ONLY.

Purpose: guard the torch-CPU oracle against "torch-vs-torch" circularity on small shapes.
There is no reference source to cite (/root/reference/README.md:1 is the whole tree);
each function states the schema whose arithmetic it restates.  Pure loops / einsum, small
inputs only.
"""
from __future__ import annotations

import numpy as np


def reflection_pad2d(x: np.ndarray, p: int) -> np.ndarray:
    """aten::reflection_pad2d — mirror without repeating the edge pixel (needs p < dim)."""
    B, C, H, W = x.shape
    hi = np.array([abs(i) if i < H else 2 * (H - 1) - i for i in range(-p, H + p)])
    wi = np.array([abs(j) if j < W else 2 * (W - 1) - j for j in range(-p, W + p)])
    return x[:, :, hi][:, :, :, wi]


def conv2d(x, w, b=None, stride=1, pad=0):
    """aten::convolution(transposed=False): cross-correlation, zero padding, NCHW / OIHW."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    B, C, H, W = x.shape
    O, _, kH, kW = w.shape
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad)); xp[:, :, pad:pad + H, pad:pad + W] = x
    Ho = (H + 2 * pad - kH) // stride + 1; Wo = (W + 2 * pad - kW) // stride + 1
    y = np.zeros((B, O, Ho, Wo))
    for kh in range(kH):
        for kw in range(kW):
            patch = xp[:, :, kh:kh + stride * (Ho - 1) + 1:stride, kw:kw + stride * (Wo - 1) + 1:stride]
            y += np.einsum("bchw,oc->bohw", patch, w[:, :, kh, kw])
    if b is not None:
        y += np.asarray(b, np.float64)[None, :, None, None]
    return y


def conv_transpose2d(x, w, b=None, stride=2, pad=1, output_padding=1):
    """aten::convolution(transposed=True): scatter form, weight (Cin, Cout, kH, kW)."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    B, C, H, W = x.shape
    _, O, kH, kW = w.shape
    Ho = (H - 1) * stride - 2 * pad + kH + output_padding
    Wo = (W - 1) * stride - 2 * pad + kW + output_padding
    full = np.zeros((B, O, (H - 1) * stride + kH + output_padding, (W - 1) * stride + kW + output_padding))
    for kh in range(kH):
        for kw in range(kW):
            full[:, :, kh:kh + stride * (H - 1) + 1:stride, kw:kw + stride * (W - 1) + 1:stride] += \
                np.einsum("bchw,co->bohw", x, w[:, :, kh, kw])
    y = full[:, :, pad:pad + Ho, pad:pad + Wo]
    if b is not None:
        y = y + np.asarray(b, np.float64)[None, :, None, None]
    return y


def instance_norm(x, eps=1e-5):
    """aten::instance_norm(use_input_stats=True, weight=None): biased variance per (n, c) plane."""
    x = np.asarray(x, np.float64)
    mu = x.mean(axis=(2, 3), keepdims=True); var = x.var(axis=(2, 3), keepdims=True)
    return (x - mu) / np.sqrt(var + eps)


def instance_norm_bwd(dy, x, eps=1e-5):
    """Gradient of instance_norm wrt x (closed form of aten::native_batch_norm_backward on the (1,B*C,H,W) view)."""
    dy = np.asarray(dy, np.float64); x = np.asarray(x, np.float64)
    mu = x.mean(axis=(2, 3), keepdims=True); var = x.var(axis=(2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps); xh = (x - mu) * rstd
    return rstd * (dy - dy.mean(axis=(2, 3), keepdims=True) - xh * (dy * xh).mean(axis=(2, 3), keepdims=True))


def relu(x): return np.maximum(x, 0.0)
def leaky_relu(x, s=0.2): return np.where(x > 0, x, s * x)
def l1_loss(a, b): return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).mean()
def mse_const(a, t): return ((np.asarray(a, np.float64) - t) ** 2).mean()


def conv2d_bwd(dy, x, w, stride=1, pad=0):
    """aten::convolution_backward(transposed=False) -> (dx, dw, db), by direct accumulation."""
    dy = np.asarray(dy, np.float64); x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    B, C, H, W = x.shape; O, _, kH, kW = w.shape; _, _, Ho, Wo = dy.shape
    xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad)); xp[:, :, pad:pad + H, pad:pad + W] = x
    dxp = np.zeros_like(xp); dw = np.zeros_like(w)
    for kh in range(kH):
        for kw in range(kW):
            sl = (slice(None), slice(None), slice(kh, kh + stride * (Ho - 1) + 1, stride), slice(kw, kw + stride * (Wo - 1) + 1, stride))
            dw[:, :, kh, kw] = np.einsum("bohw,bchw->oc", dy, xp[sl])
            dxp[sl] += np.einsum("bohw,oc->bchw", dy, w[:, :, kh, kw])
    return dxp[:, :, pad:pad + H, pad:pad + W], dw, dy.sum(axis=(0, 2, 3))


def reflection_pad2d_bwd(dyp, p):
    """aten::reflection_pad2d_backward: fold the mirrored border back onto the interior."""
    dyp = np.asarray(dyp, np.float64)
    B, C, Hp, Wp = dyp.shape; H, W = Hp - 2 * p, Wp - 2 * p
    hi = [abs(i) if i < H else 2 * (H - 1) - i for i in range(-p, H + p)]
    wi = [abs(j) if j < W else 2 * (W - 1) - j for j in range(-p, W + p)]
    tmp = np.zeros((B, C, H, Wp))
    for i, h in enumerate(hi): tmp[:, :, h] += dyp[:, :, i]
    dx = np.zeros((B, C, H, W))
    for j, ww in enumerate(wi): dx[:, :, :, ww] += tmp[:, :, :, j]
    return dx


def adam_step(p, g, m, v, step, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
    """torch.optim.Adam (amsgrad=False, wd=0) single step; returns (p, m, v)."""
    m = b1 * m + (1 - b1) * g; v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step; bc2 = 1 - b2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
    return p, m, v
