"""CPU checks of the input pipeline: the oracle's resize is pinned against the Pillow build in
this image (the library whose algorithm it restates), the product's host-side coefficient tables against the oracle's,
and the dataset / parameter-sampling host logic on a throw-away folder tree.  No device work here."""
import os

import numpy as np
import pytest

from oracle import pipeline_ref as P

SIZES = [(256, 256, 286, 286), (64, 64, 72, 72), (300, 400, 286, 286), (700, 512, 286, 286), (33, 47, 40, 40),
         (286, 286, 286, 286), (100, 120, 286, 64)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_oracle_resize_matches_pillow_bit_exact(h, w, oh, ow):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(P.resize_bicubic_u8(img, oh, ow), ref)


@pytest.mark.parametrize("h,w,oh,ow", [(256, 256, 286, 286), (300, 400, 286, 286), (33, 47, 40, 40)])
def test_oracle_bilinear_matches_pillow_bit_exact(h, w, oh, ow):
    Image = pytest.importorskip("PIL.Image")
    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(P.resize_bicubic_u8(img, oh, ow, "bilinear"), ref)
    import unpaired_image_generation_amd.pipeline as pl
    for n_in, n_out in ((h, oh), (w, ow)):
        b, k = pl.resample_tables(n_in, n_out, "bilinear")
        bo, ko = P.precompute_coeffs(n_in, n_out, "bilinear")
        assert np.array_equal(b, bo) and np.array_equal(k, ko)


def test_oracle_resize_edge_images():
    """constant, black/white checker (overshoot must clip at 0 / 255) and a 1-pixel-wide stripe"""
    Image = pytest.importorskip("PIL.Image")
    chk = (np.indices((40, 40)).sum(0) % 2 * 255).astype(np.uint8)[..., None].repeat(3, 2)
    stripe = np.zeros((37, 53, 3), np.uint8); stripe[:, 26] = 255
    for img in (np.full((31, 45, 3), 200, np.uint8), chk, stripe):
        for size in ((286, 286), (20, 24)):
            ref = np.asarray(Image.fromarray(img).resize((size[1], size[0]), Image.BICUBIC))
            assert np.array_equal(P.resize_bicubic_u8(img, *size), ref)


def test_normalize_matches_torch_chain():
    import torch
    v = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    t = torch.from_numpy(v).float().div(255)                     # ToTensor
    ref = ((t - 0.5) / 0.5).numpy()                              # Normalize(0.5, 0.5)
    out = P.to_tensor_normalize(v)
    assert out.dtype == np.float32 and np.array_equal(out, ref)
    assert out.min() == -1.0 and out.max() == 1.0


@pytest.mark.parametrize("n_in,n_out", [(256, 286), (700, 286), (286, 286), (47, 40), (33, 286), (1024, 286)])
def test_product_tables_equal_oracle(n_in, n_out):
    import unpaired_image_generation_amd.pipeline as pl
    b, k = pl.resample_tables(n_in, n_out)
    bo, ko = P.precompute_coeffs(n_in, n_out)
    assert np.array_equal(b, bo) and np.array_equal(k, ko)
    assert k.dtype == np.int32 and b.dtype == np.int32
    # every row of taps sums to one (in 22-bit fixed point, up to the per-tap rounding)
    assert np.all(np.abs(k.sum(1) - (1 << 22)) <= k.shape[1])
    assert np.all(b[:, 0] >= 0) and np.all(b[:, 0] + b[:, 1] <= n_in) and np.all(b[:, 1] <= k.shape[1])


def test_param_sampling_ranges_and_determinism():
    import torch
    import unpaired_image_generation_amd.pipeline as pl
    a = pl.DeviceAugment(286, 256, True, torch.bfloat16, "cpu", seed=3, rank=1)
    b = pl.DeviceAugment(286, 256, True, torch.bfloat16, "cpu", seed=3, rank=1)
    c = pl.DeviceAugment(286, 256, True, torch.bfloat16, "cpu", seed=3, rank=2)
    pa, pb, pc = a.sample_params(512), b.sample_params(512), c.sample_params(512)
    assert np.array_equal(pa, pb) and not np.array_equal(pa, pc)
    assert pa[:, :2].min() == 0 and pa[:, :2].max() == 30 and set(np.unique(pa[:, 2])) == {0, 1}
    noflip = pl.DeviceAugment(286, 256, False, torch.bfloat16, "cpu").sample_params(64)
    assert not noflip[:, 2].any()
    with pytest.raises(ValueError):
        pl.DeviceAugment(200, 256)


def test_unpaired_folders(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    import unpaired_image_generation_amd.pipeline as pl
    rng = np.random.default_rng(0)
    for d, n in (("trainA", 5), ("trainB", 3)):
        os.makedirs(tmp_path / d)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)).save(tmp_path / d / f"{i:02d}.png")
    (tmp_path / "trainA" / "notes.txt").write_text("not an image")
    ds = pl.UnpairedFolders(str(tmp_path), "train", serial_batches=True)
    assert len(ds) == 5 and len(ds.A) == 5 and len(ds.B) == 3
    a, b = ds.pair(4)
    assert a.endswith("trainA/04.png") and b.endswith("trainB/01.png")
    ds2 = pl.UnpairedFolders(str(tmp_path), "train", seed=7)
    ds3 = pl.UnpairedFolders(str(tmp_path), "train", seed=7)
    assert [ds2.pair(i) for i in range(10)] == [ds3.pair(i) for i in range(10)]
    assert pl.decode_rgb(a).shape == (20, 24, 3)
    with pytest.raises(FileNotFoundError):
        pl.UnpairedFolders(str(tmp_path), "test")


def test_oracle_matches_committed_golden():
    """tests/golden/pipeline_64.npz (tests/golden/make_golden.py::pipeline_small): the oracle reproduces its committed outputs
    and the committed Pillow-resized image"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_64.npz"))
    out = np.stack([P.augment(im, 72, 64, int(p[0]), int(p[1]), bool(p[2])) for im, p in zip(g["imgs"], g["params"])])
    assert np.array_equal(out, g["out"])
    assert np.array_equal(P.resize_bicubic_u8(g["small"], 36, 30), g["small_resized_36x30"])
