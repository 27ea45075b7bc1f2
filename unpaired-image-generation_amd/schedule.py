"""Host-side training utilities next to the hot path: the image history pool that feeds the
discriminators' fake batch, and the learning-rate schedule of the CycleGAN recipe [PAPER].  No dependency on the HIP
library: both are covered by CPU tests."""
from __future__ import annotations

import random

import torch


def linear_decay_scale(epoch: int, n_const: int = 100, n_decay: int = 100) -> float:
    """LR multiplier of the CycleGAN recipe: 1.0 for the first `n_const` epochs, then linearly to zero over the next
    `n_decay` epochs (the lambda of the usual LambdaLR: 1 - max(0, epoch - n_const) / (n_decay + 1))."""
    return 1.0 - max(0, epoch - n_const) / float(n_decay + 1)


class ImagePool:
    """History buffer of generated images [PAPER: Shrivastava et al. trick, pool of 50]: the discriminators are updated with
    a mix of current and past fakes.  query(images) returns a batch of the same shape: while the pool is filling, every
    image is stored and returned; afterwards each image is, with probability 1/2, swapped with a random stored one (the old
    image is returned, the new one stored), otherwise returned as is.
    Device-side: the pool is ONE preallocated tensor on the images' device; the random decisions come from a host RNG
    (seeded per rank), so there is no device synchronisation.  pool_size = 0 disables the pool (query is the identity)."""

    def __init__(self, pool_size: int = 50, seed: int = 0):
        self.size = int(pool_size)
        self.rng = random.Random(seed)
        self.buf = None
        self.n = 0

    def query(self, images: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """`out` (optional, same shape) receives the result: a static buffer a captured graph reads from."""
        if self.size == 0:
            if out is None:
                return images
            out.copy_(images)
            return out
        if self.buf is None:
            self.buf = torch.empty((self.size,) + tuple(images.shape[1:]), device=images.device, dtype=images.dtype)
        if out is None:
            out = torch.empty_like(images)
        for i in range(images.shape[0]):
            img = images[i]
            if self.n < self.size:
                self.buf[self.n].copy_(img)
                self.n += 1
                out[i].copy_(img)
            elif self.rng.random() > 0.5:
                j = self.rng.randrange(self.size)
                out[i].copy_(self.buf[j])
                self.buf[j].copy_(img)
            else:
                out[i].copy_(img)
        return out

    def state_dict(self):
        return {"size": self.size, "n": self.n, "buf": None if self.buf is None else self.buf.clone(), "rng": self.rng.getstate()}

    def load_state_dict(self, sd):
        self.size, self.n = int(sd["size"]), int(sd["n"])
        self.buf = None if sd["buf"] is None else sd["buf"].clone()
        self.rng.setstate(sd["rng"])
