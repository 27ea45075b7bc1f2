"""Generate the committed golden fixtures from the CPU oracle (stock torch ops; no reference source exists).

Run from the repo root:  python tests/golden/make_golden.py
Fixtures are DATA (inputs + expected outputs); weights are regenerated from the seed, not stored.
"""
import os, sys, json
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.torch_oracle import CycleGANOracle, Discriminator, Generator, init_weights  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def config1():
    torch.manual_seed(1234)
    g = init_weights(Generator(n_blocks=6))
    x = torch.rand(1, 3, 64, 64) * 2 - 1
    with torch.no_grad():
        y = g(x)
    np.savez_compressed(os.path.join(OUT, "config1_g6_64.npz"), x=x.numpy(), y=y.numpy())


def disc64():
    torch.manual_seed(4321)
    d = init_weights(Discriminator())
    x = torch.rand(2, 3, 64, 64) * 2 - 1
    with torch.no_grad():
        y = d(x)
    np.savez_compressed(os.path.join(OUT, "disc_64.npz"), x=x.numpy(), y=y.numpy())


def train_step_small():
    """One full §3.1 step at B=2, 64x64, G(6): 8 losses + a few post-step weight probes."""
    torch.manual_seed(7)
    o = CycleGANOracle(n_blocks=6)
    rA = torch.rand(2, 3, 64, 64) * 2 - 1
    rB = torch.rand(2, 3, 64, 64) * 2 - 1
    losses = [o.train_step(rA, rB) for _ in range(2)]
    sd = o.G_A.state_dict(); dd = o.D_A.state_dict()
    np.savez_compressed(os.path.join(OUT, "train_step_64.npz"), real_A=rA.numpy(), real_B=rB.numpy(),
                        fake_B=o.last["fake_B"].numpy(),
                        gA_1_weight=sd["1.weight"].numpy(), gA_10_b1_weight_slice=sd["10.b.1.weight"][:8, :8].numpy(),
                        gA_last_weight=sd["%d.weight" % (len(o.G_A) - 2)].numpy(), dA_0_weight=dd["0.weight"].numpy(),
                        dA_11_weight_slice=dd["11.weight"][:, :64].numpy())
    with open(os.path.join(OUT, "train_step_64_losses.json"), "w") as f:
        json.dump(losses, f, indent=1)


def train_step_256():
    with open(os.path.join(OUT, "train_step_256_losses.json"), "w") as f:
        json.dump({"seed": 0, "B": 1, "H": 256, "source": "SURVEY.md Appendix B recipe B2 (re-verified)",
                   "losses": dict(idt_A=3.4785473346710205, idt_B=3.214613437652588, G_A=1.4040476083755493,
                                  G_B=1.788599967956543, cyc_A=6.421139240264893, cyc_B=6.964799880981445,
                                  D_A=1.9957597255706787, D_B=1.6433069705963135)}, f, indent=1)


def pipeline_small():
    """Input-pipeline tail (oracle/pipeline_ref.py, itself pinned against Pillow): two 40x52 images -> resize 72 -> crop 64
    -> flip -> [-1, 1], and the resized 8-bit image of a 24x20 source produced by Pillow's own Image.resize (a second pin)."""
    from PIL import Image
    from oracle import pipeline_ref as P
    rng = np.random.default_rng(2026)
    imgs = rng.integers(0, 256, (2, 40, 52, 3), dtype=np.uint8)
    params = np.array([[0, 8, 0], [5, 2, 1]], np.int32)
    out = np.stack([P.augment(im, 72, 64, int(p[0]), int(p[1]), bool(p[2])) for im, p in zip(imgs, params)])
    small = rng.integers(0, 256, (24, 20, 3), dtype=np.uint8)
    pil = np.asarray(Image.fromarray(small).resize((30, 36), Image.BICUBIC))
    np.savez_compressed(os.path.join(OUT, "pipeline_64.npz"), imgs=imgs, params=params, out=out, small=small, small_resized_36x30=pil)


if __name__ == "__main__":
    pipeline_small()
    config1(); disc64(); train_step_small(); train_step_256()
    print(sorted(os.listdir(OUT)))
