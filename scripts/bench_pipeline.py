"""Throughput of the §8(f) rows 3-4 paths: device augment kernel, end-to-end loader (decode threads + upload + augment),
the same recipe on the host with Pillow + numpy for comparison, and generator-only inference latency.
  python scripts/bench_pipeline.py [--workers 16]"""
import argparse, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unpaired_image_generation_amd as u
import unpaired_image_generation_amd.pipeline as pl
from unpaired_image_generation_amd.inference import Translator
from PIL import Image

ap = argparse.ArgumentParser(); ap.add_argument("--workers", type=int, default=16); a = ap.parse_args()


def ev_time(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


rng = np.random.default_rng(0)
for B, hs, ws in ((4, 256, 256), (64, 256, 256), (16, 600, 800)):
    src = torch.from_numpy(rng.integers(0, 256, (B, hs, ws, 3), dtype=np.uint8)).cuda()
    aug = pl.DeviceAugment(286, 256, True, torch.bfloat16)
    params = torch.from_numpy(aug.sample_params(B)).cuda()
    out = torch.empty((B, 256, 256, 8), dtype=torch.bfloat16, device="cuda")
    ms = ev_time(lambda: aug._launch(src, params, out))
    print(f"augment kernel  B={B:3d} {hs}x{ws} -> 286 -> 256: {ms*1e3:8.1f} us  = {B/ms*1e3:10.0f} img/s", flush=True)

# host recipe (1 core): Pillow resize + crop + flip + normalise, the stock CPU pipeline's per-image work after decode
img = Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8))
t = time.time(); n = 200
for i in range(n):
    r = np.asarray(img.resize((286, 286), Image.BICUBIC))[7:263, 9:265][:, ::-1]
    x = (r.astype(np.float32) / 255 - 0.5) / 0.5
print(f"host Pillow+numpy recipe, 1 core: {n/(time.time()-t):8.0f} img/s", flush=True)

with tempfile.TemporaryDirectory() as d:
    for sub in ("trainA", "trainB"):
        os.makedirs(os.path.join(d, sub))
        for i in range(256):
            # smooth content so the JPEGs have a realistic size / decode cost
            base = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
            Image.fromarray(base).resize((256, 256), Image.BICUBIC).save(os.path.join(d, sub, f"{i:04d}.jpg"), quality=90)
    ds = pl.UnpairedFolders(d)
    for workers in sorted({1, 4, a.workers}):
        ld = pl.UnpairedLoader(ds, 4, workers=workers, prefetch=4)
        for _ in ld: pass                                     # page cache, thread start-up
        torch.cuda.synchronize(); t = time.time(); nb = 0
        for ep in range(3):
            ld.set_epoch(ep)
            for xa, xb in ld: nb += 1
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"loader end-to-end (JPEG 256x256, batch 4, {workers:2d} decode threads): {nb*8/dt:8.0f} img/s ({nb*4/dt:.0f} pairs/s)", flush=True)

for dtype in (torch.bfloat16, torch.float32):
    g = u.Generator(n_blocks=9, dtype=dtype)
    for B, H, W in ((1, 256, 256), (1, 512, 512), (8, 256, 256)):
        x = torch.rand(B, H, W, 8, device="cuda").to(dtype)
        for graph in (False, True):
            tr = Translator(g, use_graph=graph)
            ms = ev_time(lambda: tr.run_phys(x), iters=30)
            print(f"inference G9 {str(dtype)[6:]:8s} B={B} {H}x{W} graph={int(graph)}: {ms:7.3f} ms  = {B/ms*1e3:8.0f} img/s", flush=True)
