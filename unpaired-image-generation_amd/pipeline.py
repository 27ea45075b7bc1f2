"""Input pipeline: unpaired two-folder dataset -> decode on host threads -> device-side
resize(286, bicubic) + random crop(256) + random flip + scale to [-1, 1], written directly as the NHWC/8-channel tensor
the stem convolution reads (`CycleGAN.train_step` accepts it as is: no NCHW fp32 intermediate, no layout kernel).

Split of work: JPEG/PNG decode stays on the host (Pillow, which releases the GIL: a thread pool scales it); everything
after the decoded bytes is one HIP launch per domain and batch (`uig_resize_crop_flip_normalize`, csrc/augment.hip).
The resize follows Pillow's 8-bit resampling convention bit for bit (the upstream recipe runs torchvision's Resize on
PIL images), so a model trained behind this pipeline sees the same pixels as one trained behind the stock CPU pipeline;
only the crop/flip random stream is this module's own (numpy Generator seeded by (seed, rank)).
"""
from __future__ import annotations

import functools
import math
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import lib as L
from . import ops

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".tif", ".tiff", ".webp")
_COEF_BITS = 22                       # Pillow's PRECISION_BITS for 8-bit channels


def _cubic(x: float) -> float:        # Keys cubic convolution kernel, a = -0.5 (Pillow's BICUBIC), support 2
    x = abs(x)
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * -0.5
    return 0.0


def _triangle(x: float) -> float:     # Pillow's BILINEAR, support 1
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


_FILTERS = {"bicubic": (_cubic, 2.0), "bilinear": (_triangle, 1.0)}


@functools.lru_cache(maxsize=64)
def resample_tables(in_size: int, out_size: int, filt: str = "bicubic"):
    """Per output coordinate: (first source index, tap count) and the integer taps, in Pillow's convention
    (window = filter support widened by the down-scaling factor, taps normalised in float64 in source order, then
    rounded half away from zero at 22 fractional bits).  Returns (bounds int32[out,2], taps int32[out,ksize])."""
    fn, support = _FILTERS[filt]
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support *= fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    one = float(1 << _COEF_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        w = [fn((lo + i - center + 0.5) / fscale) for i in range(n)]
        total = 0.0
        for v in w:                                   # sequential sum: the order is part of the result
            total += v
        for i, v in enumerate(w):
            if total != 0.0:
                v = v / total
            taps[o, i] = int(v * one - 0.5) if v < 0 else int(v * one + 0.5)
        bounds[o] = (lo, n)
    return bounds, taps


class DeviceAugment:
    """resize(load_size) -> random crop(crop_size) -> random horizontal flip -> [-1, 1], on the device.

    `__call__(src)`: src = uint8 (B, Hs, Ws, 3) device tensor, or a list of (Hs_i, Ws_i, 3) device tensors of differing
    sizes (one launch per distinct size).  Returns the physical (B, crop, crop, 8) tensor in `dtype`."""

    def __init__(self, load_size=286, crop_size=256, flip=True, dtype=torch.bfloat16, device="cuda", seed=0, rank=0,
                 filt="bicubic"):
        if crop_size > load_size:
            raise ValueError(f"crop_size {crop_size} > load_size {load_size}")
        if filt not in _FILTERS:
            raise ValueError(f"unknown filter {filt!r}")
        self.load_size, self.crop_size, self.flip, self.dtype, self.filt = load_size, crop_size, flip, dtype, filt
        self.device = torch.device(device)
        self.rng = np.random.default_rng([seed, rank])
        self._tables = {}
        self.last_params = None

    def sample_params(self, B: int) -> np.ndarray:
        """(x0, y0, flip) per sample: uniform crop origin in [0, load - crop], flip with probability 1/2"""
        span = self.load_size - self.crop_size
        p = np.zeros((B, 3), np.int32)
        p[:, 0] = self.rng.integers(0, span + 1, B)
        p[:, 1] = self.rng.integers(0, span + 1, B)
        if self.flip:
            p[:, 2] = self.rng.random(B) > 0.5
        return p

    def _device_tables(self, Hs, Ws):
        key = (Hs, Ws)
        if key not in self._tables:
            bh, kh = resample_tables(Ws, self.load_size, self.filt)
            bv, kv = resample_tables(Hs, self.load_size, self.filt)
            self._tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in (kh, bh, kv, bv))
        return self._tables[key]

    def _launch(self, src, params_dev, out):
        B, Hs, Ws, C = src.shape
        if C != 3 or src.dtype != torch.uint8 or not src.is_contiguous() or src.device.type != "cuda":
            raise ValueError(f"expected contiguous uint8 (B,H,W,3) on the GPU, got {src.dtype} {tuple(src.shape)} on {src.device}")
        kh, bh, kv, bv = self._device_tables(Hs, Ws)
        L.check(L.lib().uig_resize_crop_flip_normalize(
            src.data_ptr(), B, Hs, Ws, kh.data_ptr(), bh.data_ptr(), kh.shape[1], kv.data_ptr(), bv.data_ptr(), kv.shape[1],
            self.load_size, self.load_size, params_dev.data_ptr(), out.data_ptr(), self.crop_size, self.crop_size,
            L.BF16 if self.dtype == torch.bfloat16 else L.F32, ops._stream()), "uig_resize_crop_flip_normalize")

    def __call__(self, src, params=None):
        items = [src] if torch.is_tensor(src) else list(src)
        B = sum(t.shape[0] if t.dim() == 4 else 1 for t in items)
        params = self.sample_params(B) if params is None else np.ascontiguousarray(params, dtype=np.int32).reshape(B, 3)
        self.last_params = params
        pd = torch.from_numpy(params).to(self.device, non_blocking=True)
        out = torch.empty((B, self.crop_size, self.crop_size, 8), dtype=self.dtype, device=self.device)
        i = 0
        for t in items:
            t4 = t if t.dim() == 4 else t.unsqueeze(0)
            n = t4.shape[0]
            self._launch(t4, pd[i:i + n], out[i:i + n])
            i += n
        return out


def list_images(folder: str, max_size: float = float("inf")):
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"{folder} is not a directory")
    out = []
    for root, _, names in sorted(os.walk(folder, followlinks=True)):
        for n in sorted(names):
            if n.lower().endswith(IMG_EXTENSIONS):
                out.append(os.path.join(root, n))
    return out[: int(min(max_size, len(out)))]


class UnpairedFolders:
    """<root>/<phase>A and <root>/<phase>B.  Item i pairs A[i mod |A|] with a random B (or B[i mod |B|] when
    serial_batches); the length is max(|A|, |B|), as in the usual unaligned two-folder dataset."""

    def __init__(self, root, phase="train", serial_batches=False, seed=0, max_size=float("inf")):
        self.A = list_images(os.path.join(root, phase + "A"), max_size)
        self.B = list_images(os.path.join(root, phase + "B"), max_size)
        if not self.A or not self.B:
            raise RuntimeError(f"no images under {root}/{phase}A or {root}/{phase}B")
        self.serial = serial_batches
        self.rng = np.random.default_rng([seed, 0x0B])

    def __len__(self):
        return max(len(self.A), len(self.B))

    def pair(self, i):
        b = i % len(self.B) if self.serial else int(self.rng.integers(len(self.B)))
        return self.A[i % len(self.A)], self.B[b]


def decode_rgb(path: str) -> np.ndarray:
    from PIL import Image           # host-side decoder; required for file input, never a fallback for device work
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class _StagingRing:
    """Reusable pinned-host + device byte buffers for one domain's uniform-size batches: allocating pinned memory per
    batch costs more than decoding it.  A slot's host side is rewritten only after its upload has completed (event);
    its device side is only touched on the copy stream, whose order protects it."""

    def __init__(self, device, nslots):
        self.device, self.slots, self.i = device, [None] * nslots, 0

    def put(self, arrays, stream):
        shape = (len(arrays),) + arrays[0].shape
        slot = self.slots[self.i]
        if slot is None or tuple(slot[0].shape) != shape:
            slot = self.slots[self.i] = [torch.empty(shape, dtype=torch.uint8).pin_memory(),
                                         torch.empty(shape, dtype=torch.uint8, device=self.device), None]
        host, dev, done = slot
        if done is not None:
            done.synchronize()
        hn = host.numpy()
        for k, a in enumerate(arrays):
            hn[k] = a
        with torch.cuda.stream(stream):
            dev.copy_(host, non_blocking=True)
            slot[2] = torch.cuda.Event()
            slot[2].record(stream)
        self.i = (self.i + 1) % len(self.slots)
        return dev


class UnpairedLoader:
    """Iterate (real_A, real_B) physical batches.  Rank r of `world` takes every world-th batch slot of a shuffled epoch
    order (same permutation on all ranks: seeded by (seed, epoch)), decodes on `workers` threads, uploads on a copy
    stream and augments on the device; `prefetch` batches are kept in flight by a producer thread."""

    def __init__(self, dataset, batch_size, dtype=torch.bfloat16, device="cuda", load_size=286, crop_size=256, flip=True,
                 shuffle=True, seed=0, rank=0, world=1, workers=4, prefetch=2, drop_last=True):
        self.ds, self.bs, self.device = dataset, batch_size, torch.device(device)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.aug_A = DeviceAugment(load_size, crop_size, flip, dtype, device, seed, 2 * rank)
        self.aug_B = DeviceAugment(load_size, crop_size, flip, dtype, device, seed, 2 * rank + 1)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.prefetch = prefetch
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = len(self.ds) // self.world
        return per_rank // self.bs if self.drop_last else -(-per_rank // self.bs)

    def _upload(self, arrays, stream, ring):
        """list of (H,W,3) uint8 arrays -> list of device tensors.  Uniform sizes (the usual case) go through a ring of
        reusable pinned staging buffers as ONE copy; a ragged batch falls back to one pinned allocation per image."""
        if len({a.shape for a in arrays}) == 1:
            return [ring.put(arrays, stream)]
        with torch.cuda.stream(stream):
            return [torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(self.device, non_blocking=True) for a in arrays]

    def _produce(self, order, q, stop):
        copy = torch.cuda.Stream(device=self.device)
        ring_a, ring_b = _StagingRing(self.device, self.prefetch + 2), _StagingRing(self.device, self.prefetch + 2)
        try:
            for s in range(len(self)):
                if stop.is_set():
                    return
                first = (s * self.world + self.rank) * self.bs
                idx = order[first: first + self.bs]
                pairs = [self.ds.pair(int(i)) for i in idx]
                imgs = list(self.pool.map(decode_rgb, [p for ab in pairs for p in ab]))
                da, db = self._upload(imgs[0::2], copy, ring_a), self._upload(imgs[1::2], copy, ring_b)
                with torch.cuda.stream(copy):
                    xa, xb = self.aug_A(da), self.aug_B(db)
                    ev = torch.cuda.Event(); ev.record(copy)
                q.put((xa, xb, ev))
            q.put(None)
        except BaseException as e:      # surface decode / launch errors on the consumer side
            q.put(e)

    def __iter__(self):
        n = len(self.ds)
        order = np.random.default_rng([self.seed, self.epoch]).permutation(n) if self.shuffle else np.arange(n)
        q, stop = queue.Queue(maxsize=max(1, self.prefetch)), threading.Event()
        t = threading.Thread(target=self._produce, args=(order, q, stop), daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                xa, xb, ev = item
                torch.cuda.current_stream(self.device).wait_event(ev)
                xa.record_stream(torch.cuda.current_stream(self.device)); xb.record_stream(torch.cuda.current_stream(self.device))
                yield xa, xb
        finally:
            stop.set()
            while t.is_alive():          # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(timeout=0.05)
