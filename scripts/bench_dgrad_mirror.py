"""ResBlock 3x3 conv at the bench shape (256->256, 64x64 maps, 16 images, two weight sets): forward and input gradient of the
reflection-padded and the zero-padded layer, HIP-event timed inside one HIP graph of 20 launches each (no host launch cost):
what the mirror-pixel input gradient costs over the plain zero-padded one and over the forward.
Usage: python scripts/bench_dgrad_mirror.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks

dt, dev, B, g = torch.bfloat16, "cuda", 16, 8
lib = u.lib.lib()


def make_graph(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                fn()
        gr.replay(); s.synchronize()
    return gr, s, n


def time_graph(g, reps=5):
    gr, s, n = g
    with torch.cuda.stream(s):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


cases, keep = {}, []
for pm in ("reflect", "zero"):
    ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, pm, dtype=dt, device=dev) for _ in range(2)]
    for l in ls:
        l.ensure_packed()
    sp = ls[0].spec
    x = (torch.rand(B, 64, 64, 256, device=dev) * 2 - 1).to(dt)
    dy = (torch.randn(B, 64, 64, 256, device=dev) * 0.5).to(dt)
    res = (torch.randn(B, 64, 64, 256, device=dev) * 0.5).to(dt)
    pair_f = (ls[1].wp_fwd, ls[1].bias, g)
    pair_d = (ls[1].wp_dgrad, None, g)
    keep.append((ls, x, dy, res))
    cases[pm + " fwd"] = make_graph(lambda: ops.conv_forward(sp, x, ls[0].wp_fwd, ls[0].bias, pair=pair_f))
    cases[pm + " fwd+INstats"] = make_graph(lambda: ops.conv_forward(sp, x, ls[0].wp_fwd, ls[0].bias, pair=pair_f, want_in_stats=True))
    cases[pm + " dgrad"] = make_graph(lambda: ops.conv_dgrad(sp, dy, ls[0].wp_dgrad, (64, 64), pair_d))
    if pm == "reflect":
        cases[pm + " dgrad+skip"] = make_graph(lambda: ops.conv_dgrad(sp, dy, ls[0].wp_dgrad, (64, 64), pair_d, res_add=res))
        for m, nm in ((3, "no step-8 fix"), (5, "no first fix"), (7, "no fix at all")):
            lib.uig_debug_set_mirror(m)
            cases[pm + " dgrad DIAG " + nm] = make_graph(lambda: ops.conv_dgrad(sp, dy, ls[0].wp_dgrad, (64, 64), pair_d))
        lib.uig_debug_set_mirror(0)
        cases[pm + " dgrad (border GEMM)"] = make_graph(lambda: ops.conv_dgrad(sp, dy, ls[0].wp_dgrad, (64, 64), pair_d))
        cases[pm + " dgrad+skip (border GEMM)"] = make_graph(lambda: ops.conv_dgrad(sp, dy, ls[0].wp_dgrad, (64, 64), pair_d, res_add=res))
        lib.uig_debug_set_mirror(1)
    if pm == "zero":      # DIAG: the mirror kernel on the zero-padded layer's operands (wrong maths, same data as "zero dgrad")
        dxz = torch.empty(B, 64, 64, 256, device=dev, dtype=dt)
        def mk(dy=dy, ls=ls, dxz=dxz):
            u.lib.check(lib.uig_reflect3x3_dgrad_mirror(dy.data_ptr(), ls[0].wp_dgrad.data_ptr(), ls[1].wp_dgrad.data_ptr(), g, None, dxz.data_ptr(),
                                                        B, 64, 64, 256, 256, 256, u.lib.BF16, torch.cuda.current_stream().cuda_stream), "mirror")
        cases["zero-layer operands, mirror kernel"] = make_graph(mk)
        lib.uig_debug_set_mirror(7)
        cases["zero-layer operands, mirror kernel no fix"] = make_graph(mk)
        lib.uig_debug_set_mirror(1)
# interleaved rounds: clocks drift over seconds, so every case is timed in every round and the median is reported
import statistics
ts = {k: [] for k in cases}
for r in range(9):
    for k, gph in cases.items():
        ts[k].append(time_graph(gph))
for k, v in ts.items():
    print(f"{k:36s} median {statistics.median(v):6.1f} us   min {min(v):6.1f}   max {max(v):6.1f}", flush=True)
