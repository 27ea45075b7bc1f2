"""Micro-benchmarks of the hot kernels at the shapes the 256x256 batch-4 train step launches (HIP-event timed,
random data, back-to-back launches).  Usage: python scripts/bench_kernels.py [--iters 30] [--dtype bf16]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", default="")
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--strip", type=int, default=1)
ap.add_argument("--batch", type=int, default=8, help="images per launch (the fused step launches 16 = pass 1 and 8 = pass 2 at per-GPU batch 4)")
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = "cuda"
L = u.lib
L.lib().uig_debug_set_tile(args.tile)
L.lib().uig_debug_set_strip(args.strip)


def timeit(fn, iters=args.iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def conv_case(name, kind, cin, cout, k, s, p, pm, B, H, W, act=0):
    if args.only and args.only not in name:
        return
    layer = networks.ConvLayer(kind, cin, cout, k, s, p, pm, act=act, dtype=dt, device=dev)
    layer.repack()
    sp = layer.spec
    x = (torch.rand(B, H, W, sp.cin_p, device=dev) * 2 - 1).to(dt)
    y = ops.conv_forward(sp, x, layer.wp_fwd, layer.bias)
    dy = (torch.rand(*y.shape[:3], sp.cout_p, device=dev) * 2 - 1).to(dt)
    Ho, Wo = y.shape[1], y.shape[2]
    macs = B * Ho * Wo * cout * cin * k * k if kind == "conv" else B * H * W * cout * cin * k * k
    fl = 2.0 * macs
    t_f = timeit(lambda: ops.conv_forward(sp, x, layer.wp_fwd, layer.bias))
    t_d = timeit(lambda: ops.conv_dgrad(sp, dy, layer.wp_dgrad, (H, W)))
    t_w = timeit(lambda: ops.conv_wgrad(sp, x, dy))
    print(f"{name:34s} GF={fl/1e9:7.2f}  fwd {t_f:8.1f}us {fl/t_f/1e6:7.1f}TF | dgrad {t_d:8.1f}us {fl/t_d/1e6:7.1f}TF | wgrad {t_w:8.1f}us {fl/t_w/1e6:7.1f}TF", flush=True)


def in_case(name, B, H, W, C, act, res):
    if args.only and args.only not in name:
        return
    x = torch.randn(B, H, W, C, device=dev).to(dt).requires_grad_(True)
    r = torch.randn(B, H, W, C, device=dev).to(dt) if res else None
    mod = networks.InstNormAct(act, 0.2)
    y = mod(x, r)
    dy = torch.randn_like(y)
    esz = 2 if dt == torch.bfloat16 else 4
    nbytes = x.numel() * esz
    t_f = timeit(lambda: mod(x.detach(), r))
    def bwd():
        x.grad = None
        yy = mod(x, r); yy.backward(dy)
    t_fb = timeit(bwd)
    print(f"{name:34s} MB={nbytes/1e6:7.2f}  fwd {t_f:8.1f}us ({(2+ (1 if res else 0))*nbytes/t_f/1e6:6.2f} TB/s alg) | fwd+bwd {t_fb:8.1f}us", flush=True)


B2 = args.batch
conv_case("res3x3 256->256 @64", "conv", 256, 256, 3, 1, 1, "reflect", B2, 64, 64)
conv_case("res3x3 256->256 @64 B4", "conv", 256, 256, 3, 1, 1, "reflect", 4, 64, 64)
conv_case("down 64->128 s2 @256", "conv", 64, 128, 3, 2, 1, "zero", B2, 256, 256)
conv_case("down 128->256 s2 @128", "conv", 128, 256, 3, 2, 1, "zero", B2, 128, 128)
conv_case("up convT 256->128 @64", "convT", 256, 128, 3, 2, 1, "zero", B2, 64, 64)
conv_case("up convT 128->64 @128", "convT", 128, 64, 3, 2, 1, "zero", B2, 128, 128)
conv_case("stem 7x7 3->64 @256", "conv", 3, 64, 7, 1, 3, "reflect", B2, 256, 256)
conv_case("head 7x7 64->3 @256", "conv", 64, 3, 7, 1, 3, "reflect", B2, 256, 256, act=L.ACT_TANH)
conv_case("D1 4x4 3->64 s2 @256", "conv", 3, 64, 4, 2, 1, "zero", B2, 256, 256, act=L.ACT_LRELU)
conv_case("D2 4x4 64->128 s2 @128", "conv", 64, 128, 4, 2, 1, "zero", B2, 128, 128)
conv_case("D3 4x4 128->256 s2 @64", "conv", 128, 256, 4, 2, 1, "zero", B2, 64, 64)
conv_case("D4 4x4 256->512 s1 @32", "conv", 256, 512, 4, 1, 1, "zero", B2, 32, 32)
conv_case("D5 4x4 512->1 s1 @31", "conv", 512, 1, 4, 1, 1, "zero", B2, 31, 31)
in_case("IN+relu 256ch @64", B2, 64, 64, 256, L.ACT_RELU, False)
in_case("IN+res 256ch @64", B2, 64, 64, 256, L.ACT_NONE, True)
in_case("IN+relu 64ch @256", B2, 256, 256, 64, L.ACT_RELU, False)
in_case("IN+relu 128ch @128", B2, 128, 128, 128, L.ACT_RELU, False)
