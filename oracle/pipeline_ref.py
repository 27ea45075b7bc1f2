"""CPU oracle for the input-pipeline tail: resize -> crop -> flip -> ToTensor -> Normalize.

TEST INFRASTRUCTURE — not product code.  Nothing under unpaired-image-generation_amd/ imports this file.

Provenance.  The reference snapshot (/root/reference/README.md:1) holds no pipeline source; the recipe is the one
SURVEY.md §8(f) row 3 quotes from the paper: resize to 286x286 (bicubic), random 256x256 crop, random horizontal flip,
scale to [-1, 1].  Upstream CycleGAN code bases realise the resize as `torchvision.transforms.Resize(..., BICUBIC)` on a
PIL image, i.e. Pillow's `Image.resize`, a third-party dependency that is not part of the reference tree (no pinned
version there).  This file restates Pillow's published 8-bit algorithm (src/libImaging/Resample.c:
`precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`,
`bicubic_filter` with a = -0.5) in plain numpy/Python loops, followed by torchvision's `ToTensor` (x / 255 in fp32) and
`Normalize(0.5, 0.5)` ((t - 0.5) / 0.5).
PARITY UNPINNED BY THE REFERENCE (it has no fixtures for this path).  What pins it instead: bit-exact agreement with the
Pillow build installed in this image (PIL 12.2.0, `Image.resize(..., BICUBIC)`) on random images, up- and down-scaling
-> tests/test_pipeline_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    """Resample.c `bicubic_filter` (Keys cubic, a = -0.5), support 2.0"""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def bilinear_filter(x: float) -> float:
    """Resample.c `bilinear_filter` (triangle), support 1.0"""
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {"bicubic": (bicubic_filter, 2.0), "bilinear": (bilinear_filter, 1.0)}


def precompute_coeffs(in_size: int, out_size: int, filt: str = "bicubic"):
    """Resample.c `precompute_coeffs` for the full-image box (in0 = 0, in1 = in_size) + `normalize_coeffs_8bpc`.
    Returns (bounds[out][2] = (xmin, count), kk[out][ksize] int32)."""
    filter_fn, support = FILTERS[filt]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filter_fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            # (int)(+-0.5 + v * (1 << PRECISION_BITS)): C truncation toward zero
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255)          # arithmetic shift on signed ints, like the C code


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int, filt: str = "bicubic") -> np.ndarray:
    """Pillow `Image.resize((out_w, out_h), BICUBIC)` for an (H, W, C) uint8 image: horizontal pass, then vertical pass,
    8-bit intermediate.  Pillow skips a pass whose size does not change; an identity pass is exact, so it is simply run."""
    H, W, _ = img.shape
    bh, kh = precompute_coeffs(W, out_w, filt)
    bv, kv = precompute_coeffs(H, out_h, filt)
    src = img.astype(np.int64)
    tmp = np.empty((H, out_w, img.shape[2]), np.int64)
    for xx in range(out_w):
        x0, n = bh[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kh[xx, :n].astype(np.int64), axes=([1], [0]))
        tmp[:, xx, :] = _clip8(acc)
    out = np.empty((out_h, out_w, img.shape[2]), np.int64)
    for yy in range(out_h):
        y0, n = bv[yy]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kv[yy, :n].astype(np.int64), tmp[y0:y0 + n], axes=([0], [0]))
        out[yy] = _clip8(acc)
    return out.astype(np.uint8)


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """torchvision ToTensor (x / 255, fp32) + Normalize(mean 0.5, std 0.5); stays HWC here."""
    t = img_u8.astype(np.float32) / np.float32(255.0)
    return (t - np.float32(0.5)) / np.float32(0.5)


def augment(img_u8: np.ndarray, load_size: int, crop: int, x0: int, y0: int, flip: bool, filt: str = "bicubic") -> np.ndarray:
    """(H, W, 3) uint8 -> (crop, crop, 3) float32 in [-1, 1]: Resize([load, load], BICUBIC) -> crop at (x0, y0) -> flip."""
    r = resize_bicubic_u8(img_u8, load_size, load_size, filt)
    c = r[y0:y0 + crop, x0:x0 + crop]
    if flip:
        c = c[:, ::-1]
    return to_tensor_normalize(np.ascontiguousarray(c))
