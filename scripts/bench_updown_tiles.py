"""Stride-2 / transposed generator layers and the PatchGAN's 4x4 layers on the generic gather kernel, per tile-selection hook
(uig_debug_set_tile: 0 auto = 128x128 two-stage, 3 = 128x128 three-stage ring, 64 = 128x64, 256 = 128x256 three-stage), timed inside
one HIP graph of 10 launches, interleaved rounds, median.   python scripts/bench_updown_tiles.py [B=16]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib; lib = L.lib()
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def make_graph(fn, n=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                fn()
        gr.replay(); s.synchronize()
    return gr, s, n


def time_graph(g, reps=3):
    gr, s, n = g
    with torch.cuda.stream(s):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


layers = [("down1 3x3s2 64->128 @256", "conv", 64, 128, 3, 2, 1, 256), ("down2 3x3s2 128->256 @128", "conv", 128, 256, 3, 2, 1, 128),
          ("up1 convT 256->128 @64", "convT", 256, 128, 3, 2, 1, 64), ("up2 convT 128->64 @128", "convT", 128, 64, 3, 2, 1, 128),
          ("D2 4x4s2 64->128 @128", "conv", 64, 128, 4, 2, 1, 128), ("D3 4x4s2 128->256 @64", "conv", 128, 256, 4, 2, 1, 64),
          ("D4 4x4s1 256->512 @32", "conv", 256, 512, 4, 1, 1, 32)]
cases, keep, info = {}, [], {}
lib.uig_debug_set_tr2(0)          # everything on the generic kernel
for name, kind, ci, co, k, st, pd, hw in layers:
    l = networks.ConvLayer(kind, ci, co, k, st, pd, dtype=dt, device="cuda"); l.repack()
    x = (torch.rand(B, hw, hw, ci, device="cuda") * 2 - 1).to(dt)
    y = ops.conv_forward(l.spec, x, l.wp_fwd, l.bias)
    dy = torch.rand_like(y)
    keep.append((l, x, y, dy))
    macs = B * y.shape[1] * y.shape[2] * co * ci * k * k if kind == "conv" else B * hw * hw * co * ci * k * k
    mb = (x.numel() + y.numel()) * 2 / 1e6
    for tile in (0, 3, 64, 256):
        lib.uig_debug_set_tile(tile)
        cases[(name, "fwd", tile)] = make_graph(lambda l=l, x=x: ops.conv_forward(l.spec, x, l.wp_fwd, l.bias))
        cases[(name, "dgrad", tile)] = make_graph(lambda l=l, dy=dy, hw=hw: ops.conv_dgrad(l.spec, dy, l.wp_dgrad, (hw, hw)))
    info[name] = (2.0 * macs, mb)
lib.uig_debug_set_tile(0)
ts = {k: [] for k in cases}
for r in range(5):
    for k, g in cases.items():
        ts[k].append(time_graph(g))
for name, *_ in layers:
    fl, mb = info[name]
    for what in ("fwd", "dgrad"):
        row = "  ".join(f"tile {t:3d}: {statistics.median(ts[(name, what, t)]):6.1f} us" for t in (0, 3, 64, 256))
        best = min(statistics.median(ts[(name, what, t)]) for t in (0, 3, 64, 256))
        print(f"{name:28s} {what:5s} {row}   | best {fl / best / 1e6:5.0f} TF, {mb / best:5.2f} TB/s algorithmic (in + out once)", flush=True)
