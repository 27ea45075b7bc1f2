"""GPU parity tests: every HIP op (forward AND backward) through the C ABI vs the CPU oracle (stock torch ops) on the
same seeded inputs.  fp32 path: exact-f32 MFMA, tight tolerance.  bf16 path: compared with the oracle evaluated on
bf16-rounded operands (fp32 accumulate), tolerance = a few bf16 ulps of the output scale (stated per test)."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods():
    import unpaired_image_generation_amd as u
    from unpaired_image_generation_amd import ops, networks
    assert u.lib.lib().uig_device_ok() == 1, "no gfx950 device visible"
    return u, ops, networks


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def Wo_of(W, k, s, p):
    return (W + 2 * p - k) // s + 1


def _tol(dtype, ref):
    scale = float(ref.abs().max()) + 1e-6
    return (2e-5 * scale + 1e-6) if dtype == torch.float32 else (1.6e-2 * scale)


CONV_CASES = [
    # kind, cin, cout, k, s, p, pad_mode, H, W, B      (the layer shapes of Appendix A at small spatial size)
    ("conv", 3, 64, 7, 1, 3, "reflect", 20, 24, 2),     # G layer 1 (Cin=3 -> small-Cin gather)
    ("conv", 64, 128, 3, 2, 1, "zero", 24, 20, 2),      # G downsample
    ("conv", 128, 256, 3, 2, 1, "zero", 12, 16, 1),
    ("conv", 256, 256, 3, 1, 1, "reflect", 12, 10, 3),  # ResBlock conv (the 88 % shape), ragged M, padded-gradient + fold dgrad
    ("conv", 256, 256, 3, 1, 1, "reflect", 16, 16, 3),  # square map: strip kernel + direct dgrad with the border-term GEMM
    ("conv", 128, 256, 3, 1, 1, "reflect", 64, 64, 2),  # the benchmark's 64x64 map (256-pixel strip tiles, image-row wgrad kernel)
    ("conv", 128, 128, 3, 1, 1, "zero", 6, 64, 3),      # image-row wgrad kernel with zero padding, rows split unevenly
    ("convT", 256, 128, 3, 2, 1, "zero", 6, 8, 2),      # upsample
    ("convT", 128, 64, 3, 2, 1, "zero", 10, 6, 1),
    ("conv", 64, 3, 7, 1, 3, "reflect", 18, 22, 2),     # G head (Cout=3 -> BN=16 tile)
    ("conv", 3, 64, 4, 2, 1, "zero", 32, 24, 2),        # D layer 1
    ("conv", 64, 128, 4, 2, 1, "zero", 16, 16, 2),
    ("conv", 256, 512, 4, 1, 1, "zero", 9, 8, 2),       # D layer 4 (odd output 8x7)
    ("conv", 512, 1, 4, 1, 1, "zero", 8, 9, 2),         # D head (Cout=1, unpadded output)
    ("conv", 3, 64, 7, 1, 3, "reflect", 40, 256, 2),    # G stem at the benchmark's row width: bf16 -> conv_cin8 (LDS-resident weights)
    ("conv", 3, 64, 7, 1, 3, "reflect", 24, 232, 1),    # ... ragged second 128-pixel segment
    ("conv", 64, 3, 7, 1, 3, "reflect", 24, 256, 2),    # G head at the benchmark's row width: bf16 -> taps-on-N head-row kernel
    ("convT", 256, 128, 3, 2, 1, "zero", 8, 64, 2),     # G up1 at its real row width: bf16 -> phase-fused transposed kernel (conv_tr2), vs the ORACLE
    ("convT", 128, 64, 3, 2, 1, "zero", 4, 128, 1),     # G up2 at its real row width (conv_tr2)
    ("conv", 64, 128, 3, 2, 1, "zero", 16, 128, 2),     # G down1: its input gradient (dy 8x64) runs on conv_tr2
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}{c[6][0]}{c[7]}")
def test_conv_fwd_bwd(case, dtype):
    u, ops, networks = _mods()
    kind, cin, cout, k, s, p, pm, H, W, B = case
    torch.manual_seed(zlib.crc32(repr(case).encode()) % 100000)      # reproducible across processes (str hashes are salted)
    layer = networks.ConvLayer(kind, cin, cout, k, s, p, pm, dtype=dtype, device="cuda")
    w = torch.randn(layer.spec.weight_shape()) * 0.05
    b = torch.randn(cout) * 0.1
    with torch.no_grad():
        layer.weight.copy_(w); layer.bias.copy_(b)
    x = torch.rand(B, cin, H, W) * 2 - 1
    rnd = _bf if dtype == torch.bfloat16 else (lambda t: t)
    xr = rnd(x).requires_grad_(True); wr = rnd(w).requires_grad_(True); br = b.clone().requires_grad_(True)
    xin = F.pad(xr, (p, p, p, p), mode="reflect") if pm == "reflect" else xr
    if kind == "conv":
        yref = F.conv2d(xin, wr, br, s, 0 if pm == "reflect" else p)
    else:
        yref = F.conv_transpose2d(xin, wr, br, s, p, output_padding=1)
    xp = ops.to_nhwc(x.cuda(), dtype).requires_grad_(True)
    yp = layer(xp)
    if dtype == torch.bfloat16 and W >= 232 and k == 7:      # the wide-row cases exist to cover these two kernels: fail if dispatch changes
        want = u.lib.K_CIN8 if cin == 3 else u.lib.K_HEADROW
        assert u.lib.lib().uig_debug_last_conv_kernel() == want, u.lib.lib().uig_debug_last_conv_kernel()
    if dtype == torch.bfloat16 and kind == "convT" and W in (64, 128):
        assert u.lib.lib().uig_debug_last_conv_kernel() == u.lib.K_TR2
    y = ops.from_nhwc(yp, cout).cpu()
    assert y.shape == yref.shape
    assert (y - yref.detach()).abs().max() <= _tol(dtype, yref), f"fwd L-inf {(y - yref.detach()).abs().max()}"
    if yp.shape[3] > cout:
        assert float(yp[..., cout:].abs().max()) == 0.0, "padded output channels must be zero"
    # backward
    dy = torch.randn_like(yref) * 0.5
    dyr = rnd(dy)
    yref.backward(dyr)
    dyp = ops.to_nhwc(dy.cuda(), dtype, yp.shape[3])
    yp.backward(dyp)
    if dtype == torch.bfloat16 and kind == "conv" and s == 2 and k == 3 and Wo_of(W, k, s, p) in (64, 128):
        assert u.lib.lib().uig_debug_last_conv_kernel() == u.lib.K_TR2      # the stride-2 conv's input gradient ran on conv_tr2
    dx = ops.from_nhwc(xp.grad, cin).cpu()
    assert (dx - xr.grad).abs().max() <= _tol(dtype, xr.grad), f"dgrad L-inf {(dx - xr.grad).abs().max()}"
    if xp.grad.shape[3] > cin:
        assert float(xp.grad[..., cin:].abs().max()) == 0.0
    dW = layer.weight.grad.cpu()
    assert (dW - wr.grad).abs().max() <= _tol(dtype, wr.grad), f"wgrad L-inf {(dW - wr.grad).abs().max()} of {wr.grad.abs().max()}"
    db = layer.bias.grad.cpu()
    assert (db - br.grad).abs().max() <= _tol(dtype, br.grad) * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("S,cin,cout,B,group", [(16, 128, 128, 3, 0), (64, 256, 256, 2, 0), (32, 64, 128, 5, 2), (8, 128, 64, 2, 0)])
def test_reflect_dgrad_border_paths_agree(S, cin, cout, B, group, dtype):
    """3x3 reflect-pad input gradient: the in-place ring launch (default) against the padded-gradient + fold path (the plain
    definition: transposed conv onto the (S+2)x(S+2) grid, mirrored rows / columns folded back) - the same products; only the
    order of the fp32 / bf16 roundings differs"""
    u, ops, networks = _mods()
    torch.manual_seed(5 + S)
    layer = networks.ConvLayer("conv", cin, cout, 3, 1, 1, "reflect", dtype=dtype, device="cuda"); layer.repack()
    l2 = networks.ConvLayer("conv", cin, cout, 3, 1, 1, "reflect", dtype=dtype, device="cuda"); l2.repack()
    dy = (torch.randn(B, S, S, layer.spec.cout_p, device="cuda") * 0.5).to(dtype)
    pair = (l2.wp_dgrad, None, group) if group else None
    old = ops.REFLECT_DGRAD_DIRECT
    try:
        ops.REFLECT_DGRAD_DIRECT = True
        a = ops.conv_dgrad(layer.spec, dy, layer.wp_dgrad, (S, S), pair)
        ops.REFLECT_DGRAD_DIRECT = False
        b = ops.conv_dgrad(layer.spec, dy, layer.wp_dgrad, (S, S), pair)
    finally:
        ops.REFLECT_DGRAD_DIRECT = old
    torch.cuda.synchronize()
    scale = float(b.float().abs().max())
    tol = (2e-5 if dtype == torch.float32 else 1.6e-2) * scale
    assert float((a.float() - b.float()).abs().max()) <= tol, f"{float((a.float() - b.float()).abs().max())} of {scale}"
    ring = torch.zeros(S, S, dtype=torch.bool, device="cuda"); ring[1] = ring[S - 2] = True; ring[:, 1] = ring[:, S - 2] = True
    assert float((a.float() - b.float())[:, ring].abs().max()) <= tol


@pytest.mark.parametrize("H,cin,cout,B,group,res", [(64, 256, 256, 8, 0, True), (64, 256, 256, 12, 4, False), (8, 128, 256, 96, 32, True),
                                                    (12, 256, 64, 64, 0, True), (20, 128, 128, 40, 24, False)],
                         ids=["bench-single", "bench-pair-uneven", "two-tile-map", "three-tile-map-1chunk", "five-tile-map"])
def test_reflect_dgrad_mirror_pixels(H, cin, cout, B, group, res):
    """Input gradient of a reflection-padded 3x3 conv with the mirrored terms folded INSIDE the persistent strip kernel (mirror
    pixels, uig_reflect3x3_dgrad_mirror): maps of 2 tiles (top + bottom only), 3, 5 and 16 tiles per image, 1, 2 and 4 K-chunks
    (the layer's cout = the gradient launch's reduction width), one and two weight sets, with and without the fused skip gradient;
    against the CPU oracle (stock torch autograd of conv2d over F.pad(reflect), bf16-rounded operands), overall and on the ring of
    pixels that receive mirrored terms (lines 1 / H-2, columns 1 / 62); and, on the square map, against the border-GEMM form."""
    u, ops, networks = _mods()
    lib, dt = u.lib.lib(), torch.bfloat16
    W = 64
    assert lib.uig_reflect3x3_dgrad_mirror_applicable(B, H, W, cout, cin, cin, u.lib.BF16) == 1
    torch.manual_seed(300 + H)
    g = group if group else B
    ls = [networks.ConvLayer("conv", cin, cout, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    ws = [torch.randn(cout, cin, 3, 3) * 0.05 for _ in ls]
    for l, w in zip(ls, ws):
        with torch.no_grad():
            l.weight.copy_(w)
        l.ensure_packed()
    dy = torch.randn(B, cout, H, W) * 0.5
    rs = torch.randn(B, cin, H, W) * 0.5 if res else None
    xr = torch.zeros(B, cin, H, W, requires_grad=True)
    parts = [(0, g, 0)] + ([(g, B, 1)] if group else [])
    yref = torch.cat([F.conv2d(F.pad(xr[a:e], (1, 1, 1, 1), mode="reflect"), _bf(ws[i])) for a, e, i in parts])
    yref.backward(_bf(dy))
    dxref = xr.grad + (_bf(rs) if res else 0)
    dyp = ops.to_nhwc(dy.cuda(), dt)
    rsp = ops.to_nhwc(rs.cuda(), dt) if res else None
    pair = (ls[1].wp_dgrad, None, g) if group else None
    dx = ops.conv_dgrad(ls[0].spec, dyp, ls[0].wp_dgrad, (H, W), pair, res_add=rsp)
    assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK
    got = ops.from_nhwc(dx, cin).cpu()
    tol = _tol(dt, dxref)
    assert (got - dxref).abs().max() <= tol, f"L-inf {(got - dxref).abs().max()} of {dxref.abs().max()}"
    ring = torch.zeros(H, W, dtype=torch.bool); ring[1] = ring[H - 2] = True; ring[:, 1] = ring[:, W - 2] = True
    assert (got - dxref)[:, :, ring].abs().max() <= tol
    # the mean error on the ring must look like the interior's (a missing or doubled mirrored term is O(1), not a rounding)
    e_ring, e_in = float((got - dxref)[:, :, ring].abs().mean()), float((got - dxref)[:, :, ~ring].abs().mean())
    assert e_ring <= 3 * e_in + 1e-6, (e_ring, e_in)
    if H == W:
        try:
            lib.uig_debug_set_mirror(0)
            assert lib.uig_reflect3x3_dgrad_mirror_applicable(B, H, W, cout, cin, cin, u.lib.BF16) == 0
            old = ops.conv_dgrad(ls[0].spec, dyp, ls[0].wp_dgrad, (H, W), pair, res_add=rsp)
        finally:
            lib.uig_debug_set_mirror(1)
        torch.cuda.synchronize()
        assert float((dx.float() - old.float()).abs().max()) <= 1.6e-2 * float(old.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("S", [16, 64, 20])
def test_resblock_skip_gradient_fusion(S, dtype):
    """ResBlock backward: the skip path's gradient summed into the first conv's input-gradient launch (ops.SkipLink) against
    autograd's own accumulation (separate add): same sums, one rounding fewer.  S = 20 is a map the strip kernel's border
    path does not take on this layer... the fallback (in-place add) must give the same result."""
    u, ops, networks = _mods()
    torch.manual_seed(40 + S)
    blk = networks.ResBlock(128, dtype, "cuda")
    for l in blk.b:
        if hasattr(l, "repack"):
            l.repack()
    x0 = (torch.randn(3, S, S, 128, device="cuda")).to(dtype)
    dy = (torch.randn(3, S, S, 128, device="cuda") * 0.5).to(dtype)
    res = []
    old = ops.FUSE_SKIP_GRAD
    try:
        for flag in (True, False):
            ops.FUSE_SKIP_GRAD = flag
            x = x0.clone().requires_grad_(True)
            z = (x * 1.0)                       # non-leaf input, as inside a generator
            y = blk(z)
            y.backward(dy)
            res.append((x.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
            for p in blk.parameters():
                p.grad = None
    finally:
        ops.FUSE_SKIP_GRAD = old
    (ga, pa), (gb, pb) = res
    scale = float(gb.float().abs().max())
    tol = (2e-5 if dtype == torch.float32 else 1.6e-2) * scale
    assert float((ga.float() - gb.float()).abs().max()) <= tol
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)                # parameter gradients do not depend on where the sum happens


@pytest.mark.parametrize("pm", ["reflect", "zero"])
def test_wgrad_row_kernel_matches_generic(pm):
    """bf16 ResBlock weight gradient: the image-row kernel (three kw taps per staged row) against the generic split-K kernel
    on the same operands - same products, different fp32 summation order."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(11)
    layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, pm, dtype=torch.bfloat16, device="cuda")
    x = (torch.rand(5, 64, 64, 256, device="cuda") * 2 - 1).to(torch.bfloat16)
    dy = (torch.randn(5, 64, 64, 256, device="cuda") * 0.5).to(torch.bfloat16)
    try:
        lib.uig_debug_set_wgrad_rows(0)
        ref = ops.conv_wgrad(layer.spec, x, dy)
    finally:
        lib.uig_debug_set_wgrad_rows(1)
    got = ops.conv_wgrad(layer.spec, x, dy)
    acc = ref.clone()
    ops.conv_wgrad(layer.spec, x, dy, out=acc, accumulate=True)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale, f"row kernel vs generic: {float((got - ref).abs().max())} of {scale}"
    assert float((acc - 2 * ref).abs().max()) <= 4e-5 * scale


@pytest.mark.parametrize("pm,H,W", [("reflect", 24, 256), ("reflect", 10, 72), ("zero", 12, 40)])
def test_wgrad_head_kernel_matches_generic(pm, H, W):
    """bf16 weight gradient of the 7x7 64 -> 3 output conv: the image-row kernel (kw taps as operand columns, halo pixels as
    extra LDS rows) against the generic split-K kernel, single launch and paired launch"""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(13)
    layer = networks.ConvLayer("conv", 64, 3, 7, 1, 3, pm, dtype=torch.bfloat16, device="cuda")
    x = (torch.rand(5, H, W, 64, device="cuda") * 2 - 1).to(torch.bfloat16)
    dy = torch.zeros(5, H, W, 8, device="cuda", dtype=torch.bfloat16)
    dy[..., :3] = (torch.randn(5, H, W, 3, device="cuda") * 0.5).to(torch.bfloat16)
    try:
        lib.uig_debug_set_wgrad_head(0)
        ref = ops.conv_wgrad(layer.spec, x, dy)
        refs = [ops.conv_wgrad(layer.spec, x[:2], dy[:2]), ops.conv_wgrad(layer.spec, x[2:], dy[2:])]
    finally:
        lib.uig_debug_set_wgrad_head(1)
    got = ops.conv_wgrad(layer.spec, x, dy)
    parts = ops.conv_wgrad_pair_partial(layer.spec, x, dy, 2)
    gots = [ops.conv_wgrad(layer.spec, x[:2], dy[:2], partial=parts[0]), ops.conv_wgrad(layer.spec, x[2:], dy[2:], partial=parts[1])]
    torch.cuda.synchronize()
    for g_, r_ in [(got, ref)] + list(zip(gots, refs)):
        scale = float(r_.abs().max())
        assert float((g_ - r_).abs().max()) <= 2e-5 * scale, f"head kernel vs generic: {float((g_ - r_).abs().max())} of {scale}"


def test_wgrad_special_kernels_vs_oracle_at_real_widths():
    """The two specialised weight-gradient kernels against the ORACLE (stock torch autograd on bf16-rounded operands), not only
    against the generic kernel, at the row widths of the 256x256 benchmark: the 7x7 head kernel (64 -> 3 channels, 256-pixel rows,
    reflection pad 3) and the image-row kernel of the ResBlock convs (256 -> 256, 64-pixel rows, reflection pad 1); the bf16
    tolerance of every op test (1.6e-2 of the reference's scale)."""
    u, ops, networks = _mods()
    dt = torch.bfloat16
    for cin, cout, k, pad, H, W, B in ((64, 3, 7, 3, 12, 256, 2), (256, 256, 3, 1, 64, 64, 2)):
        torch.manual_seed(70 + k)
        layer = networks.ConvLayer("conv", cin, cout, k, 1, pad, "reflect", dtype=dt, device="cuda")
        x = torch.rand(B, cin, H, W) * 2 - 1
        dy = torch.randn(B, cout, H, W) * 0.5
        w = torch.zeros(cout, cin, k, k, requires_grad=True)
        F.conv2d(F.pad(_bf(x), (pad,) * 4, mode="reflect"), w).backward(_bf(dy))
        got = ops.conv_wgrad(layer.spec, ops.to_nhwc(x.cuda(), dt), ops.to_nhwc(dy.cuda(), dt, layer.spec.cout_p)).cpu()
        assert got.shape == w.grad.shape
        assert (got - w.grad).abs().max() <= _tol(dt, w.grad), f"{k}x{k}: {(got - w.grad).abs().max()} of {w.grad.abs().max()}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_multi_packer_matches_per_layer_pack(dtype):
    """one-launch tile-transposing weight packer (all layers of a generator and a discriminator) against the per-layer
    reference kernel: both kernel-side operands of every layer, bit for bit"""
    u, ops, networks = _mods()
    torch.manual_seed(21)
    nets = (networks.Generator(n_blocks=2, dtype=dtype, device="cuda"), networks.Discriminator(dtype=dtype, device="cuda"))
    layers = [l for n in nets for l in n.conv_layers()]
    with torch.no_grad():
        for l in layers:
            l.weight.normal_(0, 0.05)
    for l in layers:
        l.repack()                                   # per-layer kernel
    want = [(l.wp_fwd.clone(), l.wp_dgrad.clone()) for l in layers]
    for l in layers:
        l.wp_fwd.fill_(7.0); l.wp_dgrad.fill_(7.0)
    ops.MultiPacker(layers).run()
    torch.cuda.synchronize()
    for l, (f, d) in zip(layers, want):
        assert torch.equal(l.wp_fwd, f), f"fwd operand of {l.spec}"
        assert torch.equal(l.wp_dgrad, d), f"dgrad operand of {l.spec}"


def test_wgrad_pair_launch_matches_single():
    """both networks' ResBlock weight-gradient partials from ONE launch (uneven groups: 3 + 2 images) against one launch per
    network on the image slices"""
    u, ops, networks = _mods()
    torch.manual_seed(12)
    layer = networks.ConvLayer("conv", 128, 256, 3, 1, 1, "reflect", dtype=torch.bfloat16, device="cuda")
    x = (torch.rand(5, 64, 64, 128, device="cuda") * 2 - 1).to(torch.bfloat16)
    dy = (torch.randn(5, 64, 64, 256, device="cuda") * 0.5).to(torch.bfloat16)
    parts = ops.conv_wgrad_pair_partial(layer.spec, x, dy, 3)
    assert parts is not None, "the image-row kernel should take this shape"
    for (sl, part) in ((slice(0, 3), parts[0]), (slice(3, 5), parts[1])):
        got = ops.conv_wgrad(layer.spec, x[sl], dy[sl], partial=part)
        ref = ops.conv_wgrad(layer.spec, x[sl], dy[sl])
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-5 * scale
    # layers on the generic split-K kernel (stride-2 conv, transposed conv, 4x4 discriminator conv; fp32 and bf16)
    for kind, cin, cout, k, st, pm, H, dt in (("conv", 64, 128, 3, 2, "zero", 32, torch.bfloat16), ("convT", 128, 64, 3, 2, "zero", 16, torch.float32),
                                              ("conv", 64, 128, 4, 2, "zero", 32, torch.bfloat16), ("conv", 64, 3, 7, 1, "reflect", 24, torch.float32)):
        l2 = networks.ConvLayer(kind, cin, cout, k, st, 3 if k == 7 else 1, pm, dtype=dt, device="cuda")
        x2 = (torch.rand(5, H, H, l2.spec.cin_p, device="cuda") * 2 - 1).to(dt)
        Ho, Wo = l2.spec.out_hw(H, H)
        dy2 = (torch.randn(5, Ho, Wo, l2.spec.cout_p, device="cuda") * 0.5).to(dt)
        parts = ops.conv_wgrad_pair_partial(l2.spec, x2, dy2, 2)
        for (sl, part) in ((slice(0, 2), parts[0]), (slice(2, 5), parts[1])):
            got = ops.conv_wgrad(l2.spec, x2[sl], dy2[sl], partial=part)
            ref = ops.conv_wgrad(l2.spec, x2[sl], dy2[sl])
            torch.cuda.synchronize()
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (kind, cin, cout, k)


@pytest.mark.parametrize("swap2", [0, 1])
def test_wgrad_two_batch_launch_matches_single(swap2):
    """the generator phase's combined launch (ops.combined_pass_wgrad): pass 1's batch (3 + 2 images for networks A / B) and pass 2's
    batch (2 + 1 images, with swap2 in the order B / A) reduced into ONE pair of partial slabs, against the sum of one launch per
    network and batch"""
    u, ops, networks = _mods()
    L = u.lib
    lib = L.lib()
    torch.manual_seed(40 + swap2)
    layer = networks.ConvLayer("conv", 128, 256, 3, 1, 1, "reflect", dtype=torch.bfloat16, device="cuda")
    spec = layer.spec
    x1 = (torch.rand(5, 64, 64, 128, device="cuda") * 2 - 1).to(torch.bfloat16)
    dy1 = (torch.randn(5, 64, 64, 256, device="cuda") * 0.5).to(torch.bfloat16)
    x2 = (torch.rand(3, 64, 64, 128, device="cuda") * 2 - 1).to(torch.bfloat16)
    dy2 = (torch.randn(3, 64, 64, 256, device="cuda") * 0.5).to(torch.bfloat16)
    g1, g2 = 3, 2
    args = (64, 64, 256, 64, 64, 128, 3, 3, 1, 1)
    splits = int(lib.uig_wgrad_pair2_splits(5, g1, 3, g2, swap2, *args, L.BF16))
    assert splits > 0, "the image-row kernel should take this shape"
    per = splits * 256 * 9 * 128
    ws = torch.empty((2 * per,), device="cuda", dtype=torch.float32)
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.uig_wgrad_partial_pair2(dy1.data_ptr(), x1.data_ptr(), dy2.data_ptr(), x2.data_ptr(), ws.data_ptr(), 5, g1, 3, g2, swap2,
                                        *args, L.PAD_REFLECT, splits, L.BF16, s), "uig_wgrad_partial_pair2")
    a2, b2 = (slice(g2, 3), slice(0, g2)) if swap2 else (slice(0, g2), slice(g2, 3))      # pass 2's images of network A / B
    for part, s1, s2 in (((ws[:per], splits), slice(0, g1), a2), ((ws[per:], splits), slice(g1, 5), b2)):
        got = ops.conv_wgrad(spec, x1[s1], dy1[s1], partial=part)
        ref = ops.conv_wgrad(spec, x1[s1], dy1[s1]) + ops.conv_wgrad(spec, x2[s2], dy2[s2])
        torch.cuda.synchronize()
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("kind,cin,cout,k,st,pd,pm,hw,dtype", [("conv", 64, 128, 3, 2, 1, "zero", 32, torch.bfloat16), ("convT", 128, 64, 3, 2, 1, "zero", 16, torch.bfloat16),
                                                            ("conv", 3, 64, 7, 1, 3, "reflect", 24, torch.bfloat16), ("conv", 64, 128, 4, 2, 1, "zero", 20, torch.float32),
                                                            ("conv", 64, 3, 7, 1, 3, "reflect", 24, torch.bfloat16)],
                         ids=["down-s2", "up-convT", "stem7x7", "d4x4-f32-ragged", "head7x7-all-rows-kernel"])
@pytest.mark.parametrize("swap2", [0, 1])
def test_wgrad_two_batch_launch_generic_kernel(kind, cin, cout, k, st, pd, pm, hw, dtype, swap2):
    """uig_wgrad_partial_pair2 on the GENERIC split-K kernel (the pixel splits are divided between the two tensor pairs) and on the
    all-rows 7x7 head kernel (an image is a block's unit: the second pair is a pointer select); against the sum of one launch per
    network and batch."""
    u, ops, networks = _mods()
    L = u.lib
    lib = L.lib()
    torch.manual_seed(60 + swap2 + hw)
    layer = networks.ConvLayer(kind, cin, cout, k, st, pd, pm, dtype=dtype, device="cuda")
    spec = layer.spec
    xs, dys = [], []
    for B in (5, 3):
        x = (torch.rand(B, hw, hw, spec.cin_p, device="cuda") * 2 - 1).to(dtype)
        Ho, Wo = spec.out_hw(hw, hw)
        xs.append(x); dys.append((torch.randn(B, Ho, Wo, spec.cout_p, device="cuda") * 0.5).to(dtype))
    g1, g2 = 3, 2
    P1, Q1, Mh, Mw, Np, Hq, Wq, Cq, pmode, D0, D1 = ops._wgrad_operands(spec, xs[0], dys[0])
    P2, Q2 = ops._wgrad_operands(spec, xs[1], dys[1])[:2]
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    args = (Mh, Mw, Np, Hq, Wq, Cq, k, k, st, pd)
    splits = int(lib.uig_wgrad_pair2_splits(5, g1, 3, g2, swap2, *args, dt))
    assert splits >= 2
    per = splits * Np * k * k * Cq
    ws = torch.empty((2 * per,), device="cuda", dtype=torch.float32)
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.uig_wgrad_partial_pair2(P1.data_ptr(), Q1.data_ptr(), P2.data_ptr(), Q2.data_ptr(), ws.data_ptr(), 5, g1, 3, g2, swap2,
                                        *args, pmode, splits, dt, s), "uig_wgrad_partial_pair2")
    a2, b2 = (slice(g2, 3), slice(0, g2)) if swap2 else (slice(0, g2), slice(g2, 3))
    for part, s1, s2 in (((ws[:per], splits), slice(0, g1), a2), ((ws[per:], splits), slice(g1, 5), b2)):
        got = ops.conv_wgrad(spec, xs[0][s1], dys[0][s1], partial=part)
        ref = ops.conv_wgrad(spec, xs[0][s1], dys[0][s1]) + ops.conv_wgrad(spec, xs[1][s2], dys[1][s2])
        torch.cuda.synchronize()
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("act", ["tanh", "lrelu"])
def test_conv_epilogue_activation(act, dtype):
    u, ops, networks = _mods()
    L = u.lib
    torch.manual_seed(3)
    a = L.ACT_TANH if act == "tanh" else L.ACT_LRELU
    layer = networks.ConvLayer("conv", 64, 3 if act == "tanh" else 64, 3, 1, 1, "zero", act=a, slope=0.2, dtype=dtype, device="cuda")
    cout = layer.spec.cout
    rnd = _bf if dtype == torch.bfloat16 else (lambda t: t)
    x = torch.rand(2, 64, 10, 12) * 2 - 1
    w = layer.weight.detach().cpu() * 5
    with torch.no_grad():
        layer.weight.copy_(w)
    xr = rnd(x).requires_grad_(True); wr = rnd(w).requires_grad_(True)
    pre = F.conv2d(xr, wr, layer.bias.detach().cpu(), 1, 1)
    yref = torch.tanh(pre) if act == "tanh" else F.leaky_relu(pre, 0.2)
    xp = ops.to_nhwc(x.cuda(), dtype).requires_grad_(True)
    yp = layer(xp)
    assert (ops.from_nhwc(yp, cout).cpu() - yref.detach()).abs().max() <= _tol(dtype, yref)
    dy = torch.randn_like(yref)
    yref.backward(rnd(dy))
    yp.backward(ops.to_nhwc(dy.cuda(), dtype, yp.shape[3]))
    assert (ops.from_nhwc(xp.grad, 64).cpu() - xr.grad).abs().max() <= _tol(dtype, xr.grad) * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape,act,res", [((2, 256, 16, 12), "relu", False), ((3, 256, 8, 8), "none", True),
                                           ((2, 64, 40, 36), "relu", False), ((2, 512, 31, 31), "lrelu", False),
                                           ((1, 128, 33, 17), "lrelu", False)])
def test_instnorm_fwd_bwd(shape, act, res, dtype):
    u, ops, networks = _mods()
    L = u.lib
    torch.manual_seed(5)
    B, C, H, W = shape
    a = {"relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "none": L.ACT_NONE}[act]
    rnd = _bf if dtype == torch.bfloat16 else (lambda t: t)
    x = torch.randn(shape) * 2 + 0.5
    r = torch.randn(shape)
    xr = rnd(x).requires_grad_(True); rr = rnd(r).requires_grad_(True)
    y = F.instance_norm(xr, eps=1e-5)
    y = F.relu(y) if act == "relu" else (F.leaky_relu(y, 0.2) if act == "lrelu" else y)
    if res:
        y = y + rr
    xp = ops.to_nhwc(x.cuda(), dtype).requires_grad_(True)
    rp = ops.to_nhwc(r.cuda(), dtype).requires_grad_(True) if res else None
    mod = networks.InstNormAct(a, 0.2)
    yp = mod(xp, rp)
    tol = 2e-5 * 6 if dtype == torch.float32 else 4e-2
    assert (ops.from_nhwc(yp, C).cpu() - y.detach()).abs().max() <= tol
    dy = torch.randn(shape)
    y.backward(rnd(dy))
    yp.backward(ops.to_nhwc(dy.cuda(), dtype))
    tolb = 1e-4 if dtype == torch.float32 else 4e-2 * float(xr.grad.abs().max() + 1)
    assert (ops.from_nhwc(xp.grad, C).cpu() - xr.grad).abs().max() <= tolb
    if res:
        assert (ops.from_nhwc(rp.grad, C).cpu() - rr.grad).abs().max() <= tolb


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_losses(dtype):
    u, ops, networks = _mods()
    torch.manual_seed(9)
    rnd = _bf if dtype == torch.bfloat16 else (lambda t: t)
    a, b = torch.rand(2, 3, 20, 24) * 2 - 1, torch.rand(2, 3, 20, 24) * 2 - 1
    ar = rnd(a).requires_grad_(True)
    lref = F.l1_loss(ar, rnd(b)) * 10.0
    (lref * 0.7).backward()
    ap = ops.to_nhwc(a.cuda(), dtype).requires_grad_(True)
    l = ops.l1_loss(ap, ops.to_nhwc(b.cuda(), dtype), 10.0, a.numel())
    (l * 0.7).backward()
    assert abs(l.item() - lref.item()) < 1e-5 * max(1, abs(lref.item()))
    g = ops.from_nhwc(ap.grad, 3).cpu()
    assert (g - ar.grad).abs().max() <= (1e-9 if dtype == torch.float32 else 1e-2 * float(ar.grad.abs().max()))
    # LSGAN loss on an unpadded 1-channel patch map whose size is not a multiple of the 16-byte chunk
    p = torch.randn(3, 1, 7, 9)
    pr = rnd(p).requires_grad_(True)
    for t, wgt in ((1.0, 1.0), (0.0, 0.5)):
        pr.grad = None
        lref = F.mse_loss(pr, torch.full_like(pr, t)) * wgt
        lref.backward()
        pp = ops.to_nhwc(p.cuda(), dtype, 8)[..., :1].contiguous().requires_grad_(True)
        l = ops.mse_const(pp, t, wgt)
        l.backward()
        assert abs(l.item() - lref.item()) < 1e-5 * max(1, abs(lref.item()))
        g = pp.grad.permute(0, 3, 1, 2).float().cpu()
        assert (g - pr.grad).abs().max() <= (1e-8 if dtype == torch.float32 else 1e-2 * float(pr.grad.abs().max()))


def test_adam_flat_matches_torch():
    u, ops, networks = _mods()
    torch.manual_seed(11)
    n = 10007
    p0 = torch.randn(n)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.999), eps=1e-8)
    p = p0.cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    p2 = p0.cuda(); m2 = torch.zeros(n, device="cuda"); v2 = torch.zeros(n, device="cuda")
    st = ops.new_adam_state("cuda")
    for step in (1, 2, 3, 4):
        g = torch.randn(n) * (10.0 ** (step - 3))
        pr.grad = g.clone(); opt.step()
        ops.adam_flat(p, (g * 2).cuda(), m, v, 2e-4, 0.5, 0.999, 1e-8, step, 0.5)     # grad_scale folds 1/world
        ops.adam_flat_graph(p2, g.cuda(), m2, v2, 2e-4, 0.5, 0.999, 1e-8, st, 1.0)
        assert (p.cpu() - pr.detach()).abs().max() < 2e-7
        assert (p2.cpu() - pr.detach()).abs().max() < 2e-7
    assert int(st[0]) == 4


def test_layout_roundtrip_and_reflect_fold():
    u, ops, networks = _mods()
    torch.manual_seed(13)
    x = torch.randn(2, 3, 9, 11)
    xp = ops.to_nhwc(x.cuda(), torch.float32)
    assert xp.shape == (2, 9, 11, 8) and float(xp[..., 3:].abs().max()) == 0
    assert torch.equal(ops.from_nhwc(xp, 3).cpu(), x)
    xcl = x.cuda().contiguous(memory_format=torch.channels_last)        # arbitrary input strides
    assert torch.equal(ops.to_nhwc(xcl, torch.float32), xp)
    for P, H, W in ((1, 8, 10), (3, 7, 9), (3, 16, 8)):
        dyp = torch.randn(2, 16, H + 2 * P, W + 2 * P)
        ref = torch.zeros(2, 16, H, W, requires_grad=True)
        F.pad(ref, (P, P, P, P), mode="reflect").backward(dyp)
        dypp = ops.to_nhwc(dyp.cuda(), torch.float32)
        dx = torch.empty(2, H, W, 16, device="cuda")
        u.lib.check(u.lib.lib().uig_reflect_fold(dypp.data_ptr(), dx.data_ptr(), 2, H, W, 16, P, 0, torch.cuda.current_stream().cuda_stream), "fold")
        assert (ops.from_nhwc(dx, 16).cpu() - ref.grad).abs().max() < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_instnorm_bwd_colsum_partials(dtype):
    """The InstanceNorm backward emits per-block column sums of the dx it writes (the bias gradient of the conv in front
    of it).  Finalised, they must equal the column sums of dx itself, per network half of a paired batch."""
    u, ops, networks = _mods()
    torch.manual_seed(17)
    B, H, W, C = 6, 24, 20, 128
    x = (torch.randn(B, H, W, C, device="cuda") * 2 + 0.3).to(dtype)
    st = torch.stack([x.float().mean(dim=(1, 2)), 1.0 / torch.sqrt(x.float().var(dim=(1, 2), unbiased=False) + 1e-5)], dim=2).contiguous()
    dx = ops.instnorm_backward(torch.randn_like(x), x, st, u.lib.ACT_RELU, 0.0)
    cs = dx._uig_colsum
    assert cs[2] == C and cs[1] >= 1
    for img0, nimg in ((0, B), (0, 2), (2, 4)):
        got = ops._bias_grad_from_partials(cs, img0, nimg, C, None, False).cpu()
        ref = dx[img0:img0 + nimg].float().sum(dim=(0, 1, 2)).cpu()
        scale = float(dx[img0:img0 + nimg].float().abs().sum(dim=(0, 1, 2)).max())
        assert (got - ref).abs().max() <= 2e-6 * scale + 1e-6, (img0, nimg, float((got - ref).abs().max()), scale)
    acc = torch.ones(C, device="cuda")
    ops._bias_grad_from_partials(cs, 0, B, C, acc, True)
    assert (acc.cpu() - 1 - dx.float().sum(dim=(0, 1, 2)).cpu()).abs().max() <= 2e-6 * scale + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("kind,cin,cout,k,s,p,pm,H,W", [("conv", 256, 256, 3, 1, 1, "reflect", 16, 16), ("conv", 64, 128, 3, 2, 1, "zero", 32, 32),
                                                         ("convT", 256, 128, 3, 2, 1, "zero", 16, 16), ("conv", 64, 128, 4, 2, 1, "zero", 32, 48)])
def test_fused_instnorm_statistics(kind, cin, cout, k, s, p, pm, H, W, dtype):
    """conv -> InstanceNorm with the statistics accumulated in the conv epilogue (strip, direct and transposed-phase
    kernels) equals the unfused three-kernel InstanceNorm on the same conv output."""
    u, ops, networks = _mods()
    torch.manual_seed(23)
    layer = networks.ConvLayer(kind, cin, cout, k, s, p, pm, dtype=dtype, device="cuda")
    norm = networks.InstNormAct(u.lib.ACT_RELU)
    x = (torch.rand(3, H, W, cin, device="cuda") * 2 - 1).to(dtype)
    layer.emit_in_stats = True
    c1 = layer(x)
    assert hasattr(c1, "_uig_in_partial"), "fusion rule did not fire"
    y1 = norm(c1)
    layer.emit_in_stats = False
    c2 = layer(x)
    assert not hasattr(c2, "_uig_in_partial") and torch.equal(c1, c2)
    y2 = norm(c2)
    assert (y1.float() - y2.float()).abs().max() <= (2e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("kind,ci,co,hw,B,pair", [("convT", 128, 64, 128, 2, False), ("convT", 256, 128, 64, 3, False),
                                                    ("convT", 128, 64, 128, 4, True), ("convT", 256, 128, 64, 4, True),
                                                    ("conv", 64, 128, 256, 2, False), ("conv", 128, 256, 128, 2, True)])
def test_phase_fused_transposed_kernel_matches_generic(kind, ci, co, hw, B, pair):
    """conv_tr2.hip (all four output phases of a stride-2 3x3 transposed gather in one block) against the generic gather
    kernel on the same operands: ConvTranspose2d forward (with fused InstanceNorm statistics where the layer has them) and
    the input gradient of the stride-2 Conv2d, single and paired launches.  Same MFMA products, different summation
    order over taps/chunks: bf16 outputs agree to 1 ulp-level tolerance; the statistics partials to 1e-3 relative."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(7)
    dt = torch.bfloat16
    l1 = networks.ConvLayer(kind, ci, co, 3, 2, 1, dtype=dt, device="cuda"); l1.repack()
    l2 = networks.ConvLayer(kind, ci, co, 3, 2, 1, dtype=dt, device="cuda"); l2.repack()
    with torch.no_grad():
        l1.bias.normal_(); l2.bias.normal_()
    res = {}
    for mode in (0, 1):
        lib.uig_debug_set_tr2(mode)
        try:
            if kind == "convT":
                x = (torch.rand(B, hw, hw, ci, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) * 2 - 1).to(dt)
                pr = (l2.wp_fwd, l2.bias, B // 2) if pair else None
                out = ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pr, want_in_stats=True)
            else:
                ho = hw // 2
                dy = (torch.rand(B, ho, ho, co, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) * 2 - 1).to(dt)
                pr = (l2.wp_dgrad, None, B // 2) if pair else None
                out = ops.conv_dgrad(l1.spec, dy, l1.wp_dgrad, (hw, hw), pr)
            torch.cuda.synchronize()
            res[mode] = out
        finally:
            lib.uig_debug_set_tr2(1)
    ya, yb = res[0], res[1]
    assert ya.shape == yb.shape
    d = (ya.float() - yb.float()).abs()
    scale = float(ya.float().abs().max())
    assert float(d.max()) <= 0.02 * scale + 1e-3, (float(d.max()), scale)
    assert float(d.mean()) <= 2e-3 * scale
    pa, pb = getattr(ya, "_uig_in_partial", None), getattr(yb, "_uig_in_partial", None)
    if kind == "convT":            # fused InstanceNorm partials (the generic kernel has them for > 64 channels only)
        assert pb is not None and (pa is not None) == (co > 64)
        sb = pb[0].view(B, pb[1], yb.shape[3], 2).sum(1)
        yf = yb.float().reshape(B, -1, yb.shape[3])
        direct = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=2)          # statistics of the tensor as stored
        assert torch.allclose(sb, direct, rtol=1e-3, atol=1e-3 * float(direct.abs().max()))
        if pa is not None:
            assert pa[1] == pb[1]
            sa = pa[0].view(B, pa[1], ya.shape[3], 2).sum(1)
            assert torch.allclose(sa, sb, rtol=2e-3, atol=2e-3 * float(sa.abs().max()))


@pytest.mark.parametrize("nc", [2, 4])
def test_head_taps_on_n_kernel_other_channel_counts(nc):
    """the (kw * Nc + co) column packing of conv_headrow_kernel for Nc = 2 and 4 output channels, against the row-strip kernel"""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(nc)
    dt = torch.bfloat16
    l1 = networks.ConvLayer("conv", 64, nc, 7, 1, 3, "reflect", dtype=dt, device="cuda")
    with torch.no_grad():
        l1.weight.mul_(3.0); l1.bias.normal_()
    l1.repack()
    x = (torch.rand(2, 256, 256, 64, device="cuda") * 2 - 1).to(dt)
    res = {}
    for mode in (2, 1):
        lib.uig_debug_set_rowstrip(mode)
        try:
            res[mode] = ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias).float()
            torch.cuda.synchronize()
        finally:
            lib.uig_debug_set_rowstrip(1)
    scale = float(res[2].abs().max())
    assert float((res[1] - res[2]).abs().max()) <= 1.6e-2 * scale and not res[1][..., nc:].any()


@pytest.mark.parametrize("B,hw,pair,pad_mode", [(2, 256, False, "reflect"), (4, 256, True, "reflect"), (1, 512, False, "reflect"), (2, 256, False, "zero"), (1, 258, False, "reflect")])
def test_head_taps_on_n_kernel_matches_rowstrip(B, hw, pair, pad_mode):
    """conv_headrow_kernel (7x7, 64 -> 3: horizontal taps on the MFMA N side + one shift-add) against the row-strip kernel it
    replaces and against the oracle conv on bf16-rounded operands."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(3)
    dt = torch.bfloat16
    l1 = networks.ConvLayer("conv", 64, 3, 7, 1, 3, pad_mode, act=u.lib.ACT_TANH, dtype=dt, device="cuda"); l1.repack()
    l2 = networks.ConvLayer("conv", 64, 3, 7, 1, 3, pad_mode, act=u.lib.ACT_TANH, dtype=dt, device="cuda"); l2.repack()
    with torch.no_grad():
        l1.weight.mul_(3.0); l1.bias.normal_(); l2.weight.mul_(3.0); l2.bias.normal_()
    l1.repack(); l2.repack()
    x = (torch.rand(B, hw, hw, 64, device="cuda") * 2 - 1).to(dt)
    pr = (l2.wp_fwd, l2.bias, B // 2) if pair else None
    res = {}
    for mode in (2, 1):             # row-strip kernel | taps on N (default)
        lib.uig_debug_set_rowstrip(mode)
        try:
            res[mode] = ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pr)
            torch.cuda.synchronize()
        finally:
            lib.uig_debug_set_rowstrip(1)
    a, b = res[2].float(), res[1].float()
    assert a.shape == b.shape == (B, hw, hw, 8)
    assert float((a - b).abs().max()) <= 1.6e-2 and not b[..., 3:].any()
    # oracle on the first image (fp32 conv of the bf16-rounded operands)
    xi = x[:1, :, :, :].float().permute(0, 3, 1, 2).cpu()
    w = l1.weight.detach().to(dt).float().cpu()
    xp = F.pad(xi, (3, 3, 3, 3), mode="reflect") if pad_mode == "reflect" else F.pad(xi, (3, 3, 3, 3))
    ref = torch.tanh(F.conv2d(xp, w, l1.bias.detach().float().cpu()))
    got = b[:1, :, :, :3].permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max()) <= 1.6e-2


@pytest.mark.parametrize("H,W", [(37, 50), (8, 266), (40, 512), (9, 7)])
def test_taps_on_n_kernel_odd_sizes_forward_and_padded_gradient(H, W):
    """conv_headrow_kernel at ragged sizes (row groups of 4 that do not divide H, one segment of up to 266 pixels or 256-pixel
    segments): the 64 -> 3 forward with reflection padding and the stem's padded input gradient (3 <- 64, mirrored taps),
    each against the generic kernels (hook rowstrip=0)."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(H * 1000 + W)
    dt = torch.bfloat16
    head = networks.ConvLayer("conv", 64, 3, 7, 1, 3, "reflect", dtype=dt, device="cuda")
    stem = networks.ConvLayer("conv", 3, 64, 7, 1, 3, "reflect", dtype=dt, device="cuda")
    with torch.no_grad():
        head.weight.mul_(3.0); head.bias.normal_(); stem.weight.mul_(3.0)
    head.repack(); stem.repack()
    x = (torch.rand(2, H, W, 64, device="cuda") * 2 - 1).to(dt)
    res = {}
    for mode in (0, 1):
        lib.uig_debug_set_rowstrip(mode)
        try:
            res[mode] = (ops.conv_forward(head.spec, x, head.wp_fwd, head.bias).float(), ops.conv_dgrad(stem.spec, x, stem.wp_dgrad, (H, W)).float())
            torch.cuda.synchronize()
        finally:
            lib.uig_debug_set_rowstrip(1)
    for a, b in zip(res[0], res[1]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1.6e-2 * (float(a.abs().max()) + 1e-3)


@pytest.mark.parametrize("B,group", [(16, 8), (8, 0), (12, 4)], ids=["paired16", "single8", "paired12-uneven"])
def test_strip_persistent_256x128_bench_shape(B, group):
    """The instantiation the headline bench measures (conv_strip_pk_kernel<bf16,448>: 256x128 tiles, persistent blocks) at the
    bench's own shape: ResBlock 3x3 reflect conv 256->256 on 64x64 maps, paired launch over 16 images (two rounds of 256 tiles),
    one network over 8 images (one round) and an uneven pair over 12 images (half the blocks walk two tiles, and the weight set
    changes between a block's tiles).  Forward with the fused InstanceNorm statistics, input gradient with the mirrored-border
    terms and the ResBlock skip gradient summed in the epilogue, weight and bias gradients - all against the CPU oracle (stock
    torch conv2d autograd on bf16-rounded operands); tolerance 1.6e-2 * max|ref| (a few bf16 ulps of the output scale)."""
    u, ops, networks = _mods()
    lib, dt = u.lib.lib(), torch.bfloat16
    assert lib.uig_conv_strip_tile(B, 64, 64, 256, 256, 64, 64, -1, 1, u.lib.BF16) == 257
    torch.manual_seed(1000 + B)
    g = group if group else B
    ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    ws = [torch.randn(256, 256, 3, 3) * 0.03 for _ in ls]
    bs = [torch.randn(256) * 0.1 for _ in ls]
    for l, w, b in zip(ls, ws, bs):
        with torch.no_grad():
            l.weight.copy_(w); l.bias.copy_(b)
        l.emit_in_stats = True
        l.ensure_packed()
    x = torch.rand(B, 256, 64, 64) * 2 - 1
    dy = torch.randn(B, 256, 64, 64) * 0.5
    res = torch.randn(B, 256, 64, 64) * 0.5
    # ---- oracle
    xr = _bf(x).requires_grad_(True)
    wr = [_bf(w).requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    parts = [(0, g, 0)] + ([(g, B, 1)] if group else [])
    yref = torch.cat([F.conv2d(F.pad(xr[a:e], (1, 1, 1, 1), mode="reflect"), wr[i], br[i]) for a, e, i in parts])
    yref.backward(_bf(dy))
    # ---- HIP path
    xp = ops.to_nhwc(x.cuda(), dt).requires_grad_(True)
    link = ops.SkipLink()
    if group:
        yp = ops.PairConvFn.apply(xp, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, ls[0], ls[1], g, link)
    else:
        yp = ops.ConvFn.apply(xp, ls[0].weight, ls[0].bias, ls[0], link)
    assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK
    y = ops.from_nhwc(yp, 256).cpu()
    tol = _tol(dt, yref)
    assert (y - yref.detach()).abs().max() <= tol, f"fwd L-inf {(y - yref.detach()).abs().max()} (tol {tol})"
    # fused InstanceNorm statistics: the norm that consumes them against F.instance_norm of the SAME stored tensor
    assert getattr(yp, "_uig_in_partial", None) is not None
    z = ops.InstNormActFn.apply(yp, None, u.lib.ACT_RELU, 0.0, 1e-5)
    zref = F.relu(F.instance_norm(ops.from_nhwc(yp.detach(), 256).cpu(), eps=1e-5))
    assert (ops.from_nhwc(z.detach(), 256).cpu() - zref).abs().max() <= 1.6e-2 * float(zref.abs().max())
    # backward: dgrad (+ border terms + skip gradient in the epilogue), wgrad, bias grad
    link.grad = ops.to_nhwc(res.cuda(), dt)
    yp.backward(ops.to_nhwc(dy.cuda(), dt))
    assert lib.uig_conv_strip_tile(B, 64, 64, 256, 256, 64, 64, -1, 1, u.lib.BF16) == 257
    dx = ops.from_nhwc(xp.grad, 256).cpu()
    dxref = xr.grad + _bf(res)
    assert (dx - dxref).abs().max() <= _tol(dt, dxref), f"dgrad L-inf {(dx - dxref).abs().max()} of {dxref.abs().max()}"
    ring = torch.zeros(64, 64, dtype=torch.bool); ring[1] = ring[62] = True; ring[:, 1] = ring[:, 62] = True
    assert (dx - dxref)[:, :, ring].abs().max() <= _tol(dt, dxref)
    for l, w_, b_ in zip(ls, wr, br):
        assert (l.weight.grad.cpu() - w_.grad).abs().max() <= _tol(dt, w_.grad), "wgrad"
        assert (l.bias.grad.cpu() - b_.grad).abs().max() <= 2 * _tol(dt, b_.grad), "bias grad"


def test_strip_persistent_equals_tile_per_block_fp32():
    """fp32 (exact-f32 MFMA) flavour of the persistent kernel (no cross-tile prefetch: its scratch spans both LDS regions)
    against the one-tile-per-block kernel: same accumulation order, so bitwise equal; and against the oracle at 2e-5."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    torch.manual_seed(77)
    l1 = networks.ConvLayer("conv", 128, 128, 3, 1, 1, "reflect", dtype=torch.float32, device="cuda"); l1.repack()
    l2 = networks.ConvLayer("conv", 128, 128, 3, 1, 1, "reflect", dtype=torch.float32, device="cuda"); l2.repack()
    x = torch.rand(28, 128, 64, 64) * 2 - 1                      # 28 x 16 tiles = 448: one full round + a partial one
    xp = ops.to_nhwc(x.cuda(), torch.float32)
    pair = (l2.wp_fwd, l2.bias, 12)
    try:
        lib.uig_debug_set_strip(1)
        a = ops.conv_forward(l1.spec, xp, l1.wp_fwd, l1.bias, pair=pair)
        assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK
        lib.uig_debug_set_strip(3)
        b = ops.conv_forward(l1.spec, xp, l1.wp_fwd, l1.bias, pair=pair)
        assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP256
    finally:
        lib.uig_debug_set_strip(1)
    assert torch.equal(a, b)
    with torch.no_grad():
        xin = F.pad(x, (1, 1, 1, 1), mode="reflect")
        yref = torch.cat([F.conv2d(xin[:12], l1.weight.cpu(), l1.bias.cpu()), F.conv2d(xin[12:], l2.weight.cpu(), l2.bias.cpu())])
    assert (ops.from_nhwc(a, 128).cpu() - yref).abs().max() <= _tol(torch.float32, yref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 19, 30, 26, 37, 3, 1, 1, "reflect"), (1, 3, 64, 64, 64, 7, 1, 3, "reflect"), (2, 64, 33, 40, 128, 3, 2, 1, "zero")],
                         ids=["c19-37k3", "c3-64k7", "c64-128k3s2"])
def test_conv2d_fwd_nchw_boundary(case, dtype):
    """SURVEY §8(b)'s contiguous-NCHW boundary as ONE C call (uig_conv2d_fwd: repacks inside): aten::convolution on NCHW tensors,
    channel counts that are not multiples of 8 included, against F.conv2d."""
    u, ops, networks = _mods()
    lib = u.lib.lib()
    B, cin, H, W, cout, k, s, p, pm = case
    torch.manual_seed(zlib.crc32(repr(case).encode()) % 100000)
    x = torch.rand(B, cin, H, W) * 2 - 1
    w = torch.randn(cout, cin, k, k) * 0.05
    b = torch.randn(cout) * 0.1
    rnd = _bf if dtype == torch.bfloat16 else (lambda t: t)
    xin = F.pad(rnd(x), (p, p, p, p), mode="reflect") if pm == "reflect" else rnd(x)
    yref = F.conv2d(xin, rnd(w), b, s, 0 if pm == "reflect" else p)
    xd, wd, bd = x.to(dtype).cuda().contiguous(), w.cuda().contiguous(), b.cuda()
    y = torch.empty(yref.shape, device="cuda", dtype=dtype)
    dt = u.lib.BF16 if dtype == torch.bfloat16 else u.lib.F32
    n = int(lib.uig_conv2d_fwd_workspace_bytes(B, cin, H, W, cout, k, k, s, p, dt))
    ws = torch.empty(n, device="cuda", dtype=torch.uint8)
    rc = lib.uig_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, cin, H, W, cout, k, k, s, p,
                            u.lib.PAD_REFLECT if pm == "reflect" else u.lib.PAD_ZERO, dt, ws.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    u.lib.check(rc, "uig_conv2d_fwd")
    assert (y.float().cpu() - yref).abs().max() <= _tol(dtype, yref)
    assert lib.uig_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, cin, H, W, cout, k, k, s, p, 0, dt, ws.data_ptr(), 16, None) < 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 19, 30, 26, 37, 3, 1, 1, "reflect"), (1, 3, 40, 64, 64, 7, 1, 3, "reflect"), (2, 64, 33, 40, 128, 3, 2, 1, "zero"), (3, 128, 16, 64, 256, 3, 1, 1, "reflect"),
                                  (2, 64, 17, 20, 192, 4, 2, 1, "zero")],
                         ids=["odd-channels-reflect", "stem7x7", "stride2", "resblock-shape", "patchgan-k4"])
def test_conv2d_bwd_nchw_boundary(case, dtype):
    """aten::convolution_backward for Conv2d on the contiguous-NCHW C boundary (uig_conv2d_bwd: repacks, the
    input-gradient gather (+ reflection fold), split-K weight gradient + reduce and the bias column sum inside the call) against
    F.conv2d autograd on the CPU: dx, dW, db; and the output-mask forms (dx only / dW + db only)."""
    u, ops, networks = _mods()
    B, cin, H, W, cout, k, s_, p, pm = case
    torch.manual_seed(41)
    x = torch.randn(B, cin, H, W) * 0.8
    w = torch.randn(cout, cin, k, k) * 0.05
    Ho, Wo = Wo_of(H, k, s_, p), Wo_of(W, k, s_, p)
    dy = torch.randn(B, cout, Ho, Wo) * 0.5
    rd = (lambda t: _bf(t)) if dtype == torch.bfloat16 else (lambda t: t)
    xr, wr = rd(x).requires_grad_(True), rd(w).requires_grad_(True)
    br = torch.zeros(cout, requires_grad=True)
    xin = F.pad(xr, (p,) * 4, mode="reflect") if pm == "reflect" else xr
    F.conv2d(xin, wr, br, stride=s_, padding=0 if pm == "reflect" else p).backward(rd(dy))
    lib = u.lib.lib()
    dt = u.lib.BF16 if dtype == torch.bfloat16 else u.lib.F32
    n = int(lib.uig_conv2d_bwd_workspace_bytes(B, cin, H, W, cout, k, k, s_, p, dt))
    ws = torch.empty(n, device="cuda", dtype=torch.uint8)
    xd, dyd, wd = x.cuda().to(dtype).contiguous(), dy.cuda().to(dtype).contiguous(), w.cuda().contiguous()
    dx = torch.full((B, cin, H, W), float("nan"), device="cuda", dtype=dtype)
    dW = torch.full((cout, cin, k, k), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    PM = u.lib.PAD_REFLECT if pm == "reflect" else u.lib.PAD_ZERO
    st = torch.cuda.current_stream().cuda_stream
    u.lib.check(lib.uig_conv2d_bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), B, cin, H, W, cout, k, k, s_, p, PM, dt,
                                   ws.data_ptr(), n, st), "uig_conv2d_bwd")
    assert (dx.float().cpu() - xr.grad).abs().max() <= _tol(dtype, xr.grad)
    assert (dW.cpu() - wr.grad).abs().max() <= (2e-5 if dtype == torch.float32 else 2e-3) * float(wr.grad.abs().max())      # bf16 operands, fp32 accumulate
    assert (db.cpu() - br.grad).abs().max() <= 2e-5 * float(rd(dy).abs().sum((0, 2, 3)).max())
    # output masks
    dx2 = torch.empty_like(dx)
    u.lib.check(lib.uig_conv2d_bwd(dyd.data_ptr(), None, wd.data_ptr(), dx2.data_ptr(), None, None, B, cin, H, W, cout, k, k, s_, p, PM, dt, ws.data_ptr(), n, st), "uig_conv2d_bwd")
    assert torch.equal(dx2, dx)
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    u.lib.check(lib.uig_conv2d_bwd(dyd.data_ptr(), xd.data_ptr(), None, None, dW2.data_ptr(), db2.data_ptr(), B, cin, H, W, cout, k, k, s_, p, PM, dt, ws.data_ptr(), n, st), "uig_conv2d_bwd")
    assert torch.equal(dW2, dW) and torch.equal(db2, db)
    assert lib.uig_conv2d_bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), B, cin, H, W, cout, k, k, s_, p, PM, dt, ws.data_ptr(), 16, None) < 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 256, 12, 64, 128), (1, 128, 9, 14, 64), (3, 24, 10, 6, 19)], ids=["up1-row64", "up2-odd", "odd-channels"])
def test_conv_transpose2d_fwd_nchw_boundary(case, dtype):
    """aten::convolution (transposed) for ConvTranspose2d(k3, s2, p1, output_padding 1) on the contiguous-NCHW
    C boundary (uig_conv_transpose2d_fwd) against F.conv_transpose2d on the CPU."""
    u, ops, networks = _mods()
    B, cin, H, W, cout = case
    torch.manual_seed(43)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cin, cout, 3, 3) * 0.05
    b = torch.randn(cout) * 0.1
    rd = (lambda t: _bf(t)) if dtype == torch.bfloat16 else (lambda t: t)
    ref = F.conv_transpose2d(rd(x), rd(w), b, stride=2, padding=1, output_padding=1)
    lib = u.lib.lib()
    dt = u.lib.BF16 if dtype == torch.bfloat16 else u.lib.F32
    n = int(lib.uig_conv_transpose2d_fwd_workspace_bytes(B, cin, H, W, cout, dt))
    ws = torch.empty(n, device="cuda", dtype=torch.uint8)
    xd, wd, bd = x.cuda().to(dtype).contiguous(), w.cuda().contiguous(), b.cuda()
    y = torch.full((B, cout, 2 * H, 2 * W), float("nan"), device="cuda", dtype=dtype)
    u.lib.check(lib.uig_conv_transpose2d_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, cin, H, W, cout, dt, ws.data_ptr(), n,
                                             torch.cuda.current_stream().cuda_stream), "uig_conv_transpose2d_fwd")
    assert (y.float().cpu() - ref).abs().max() <= _tol(dtype, ref)
    assert lib.uig_conv_transpose2d_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, cin, H, W, cout, dt, ws.data_ptr(), 16, None) < 0


@pytest.mark.parametrize("B,group,act", [(16, 8, "relu"), (8, 0, "none"), (3, 0, "relu")], ids=["paired16-relu", "single8-none", "small3-relu"])
def test_instnorm_backward_statistics_from_dgrad_epilogue(B, group, act):
    """InstanceNorm -> 3x3 reflect conv (the ResBlock pattern): the norm's backward statistics (sum g, sum g*xhat) come out of the
    epilogue of the conv's input-gradient launch (which writes the norm's dy) instead of the norm's own pass over dy and x.
    Against (a) the same chain with that fusion off: the conv's dx bitwise equal, the norm's dx within bf16 rounding (only the
    fp32 summation order of the statistics differs); (b) the oracle (F.instance_norm + F.conv2d autograd), bf16 tolerance."""
    u, ops, networks = _mods()
    dt = torch.bfloat16
    torch.manual_seed(300 + B)
    g = group if group else B
    ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        l.ensure_packed()
    A = u.lib.ACT_RELU if act == "relu" else u.lib.ACT_NONE
    x = torch.randn(B, 256, 64, 64) * 1.5 + 0.3
    dy = torch.randn(B, 256, 64, 64) * 0.5
    res = torch.randn(B, 256, 64, 64) * 0.5

    def run(fuse):
        old = ops.FUSE_BWD_STATS
        ops.FUSE_BWD_STATS = fuse
        try:
            for l in ls:
                l.weight.grad = None; l.bias.grad = None
            xp = ops.to_nhwc(x.cuda(), dt).requires_grad_(True)
            h = ops.InstNormActFn.apply(xp * 1.0, None, A, 0.0, 1e-5)
            assert (getattr(h, "_uig_bst", None) is not None) == fuse
            link = ops.SkipLink()
            if group:
                y = ops.PairConvFn.apply(h, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, ls[0], ls[1], g, link)
            else:
                y = ops.ConvFn.apply(h, ls[0].weight, ls[0].bias, ls[0], link)
            link.grad = ops.to_nhwc(res.cuda(), dt)
            hg = []
            h.register_hook(lambda t: hg.append(t.clone()))
            y.backward(ops.to_nhwc(dy.cuda(), dt))
            return xp.grad.clone(), hg[0]
        finally:
            ops.FUSE_BWD_STATS = old

    # the fused statistics belong to the border-buffer form of the input gradient (the mirror-pixel launch does not carry them):
    # both runs on that form, so that the conv's dx can be compared bitwise
    try:
        u.lib.lib().uig_debug_set_mirror(0)
        dx1, dh1 = run(True)
        dx0, dh0 = run(False)
    finally:
        u.lib.lib().uig_debug_set_mirror(1)
    assert torch.equal(dh1, dh0), "the conv's input gradient must not change"
    sc = float(dx0.float().abs().max())
    assert float((dx1.float() - dx0.float()).abs().max()) <= 1e-2 * sc
    assert float((dx1.float() - dx0.float()).abs().mean()) <= 2e-4 * sc          # the same numbers up to a few roundings at bf16 ties
    # oracle
    xr = _bf(x).requires_grad_(True)
    hr = F.instance_norm(xr, eps=1e-5)
    hr = F.relu(hr) if act == "relu" else hr
    hb = _bf(hr.detach()).requires_grad_(True)                                   # the conv sees the bf16-rounded norm output
    parts = [(0, g, 0)] + ([(g, B, 1)] if group else [])
    yr = torch.cat([F.conv2d(F.pad(hb[a:e], (1, 1, 1, 1), mode="reflect"), _bf(ls[i].weight.detach().cpu()), ls[i].bias.detach().cpu()) for a, e, i in parts])
    yr.backward(_bf(dy))
    hr.backward(_bf(hb.grad + _bf(res)))
    ref = xr.grad
    got = ops.from_nhwc(dx1, 256).cpu()
    assert (got - ref).abs().max() <= 2.5e-2 * float(ref.abs().max()), float((got - ref).abs().max()) / float(ref.abs().max())


@pytest.mark.parametrize("B,paired", [(16, True), (8, False), (12, True), (24, True)], ids=["paired16", "single8", "paired12", "paired24-3tiles"])
def test_norm_conv_resblock_equals_apply_pass_path(B, paired, monkeypatch):
    """Round 3: the second convolution of a ResBlock applies the InstanceNorm + ReLU in front of it to its own input strip in LDS
    (ops.NormConvFn, conv_strip_pk_kernel<NORM>): no apply pass between the block's two convolutions (its finalize launch stays).  Whole ResBlocks (one
    network, and two networks in paired launches; 1, 2 and 3 tiles per persistent block) forward + backward against the path
    with the apply pass: the normalised activations are bitwise what the apply kernel writes, so the block output, the input
    gradient and every parameter gradient are bitwise equal; and the block output against the oracle (stock torch ResBlock on
    bf16-rounded weights), 1.6e-2 * max|ref|.  Also without autograd (inference: nothing stored for a backward pass)."""
    u, ops, networks = _mods()
    from oracle.torch_oracle import ResBlock as OResBlock
    dt = torch.bfloat16
    torch.manual_seed(500 + B)
    nets = [torch.nn.Sequential(networks.ResBlock(256, dt, "cuda")) for _ in range(2 if paired else 1)]
    for n in nets:
        rb = n[0]
        with torch.no_grad():
            rb.b[1].bias.normal_(0, 0.1); rb.b[5].bias.normal_(0, 0.1)
            rb.b[1].weight.mul_(1.5)
        rb.b[1].emit_in_stats = rb.b[5].emit_in_stats = True
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.randn(B, 64, 64, 256, device="cuda") * 0.5).to(dt)
    calls = []
    real = ops.norm_conv
    monkeypatch.setattr(ops, "norm_conv", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(flag, grad=True):
        monkeypatch.setattr(ops, "NORM_CONV", flag)
        for n in nets:
            for p in n.parameters():
                p.grad = None
        xp = x.clone().requires_grad_(grad)
        with (torch.enable_grad() if grad else torch.no_grad()):
            y = networks.pair_forward_phys(nets[0], nets[1], xp) if paired else nets[0][0](xp)
        if not grad:
            return (y,)
        y.backward(dy)
        return (y.detach(), xp.grad.clone(), *[p.grad.clone() for n in nets for p in n.parameters()])

    n0 = len(calls)
    fused = run(True)
    assert len(calls) == n0 + 1, "the fused launch did not run"
    assert u.lib.lib().uig_debug_last_conv_kernel() in (u.lib.K_STRIP_PK, u.lib.K_IGEMM)
    plain = run(False)
    assert len(calls) == n0 + 1
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    fused_ng = run(True, grad=False)
    assert len(calls) == n0 + 2 and torch.equal(fused_ng[0], fused[0])
    # oracle
    g = B // 2 if paired else B
    outs = []
    for i, n in enumerate(nets):
        o = OResBlock(256)
        sd = {k: (_bf(v) if k.endswith("weight") else v.clone()) for k, v in n[0].state_dict().items()}
        o.load_state_dict({k: v.cpu() for k, v in sd.items()})
        xs = ops.from_nhwc(x[:g] if i == 0 else x[g:], 256).cpu() if paired else ops.from_nhwc(x, 256).cpu()
        with torch.no_grad():
            outs.append(o(xs))
    ref = torch.cat(outs)
    got = ops.from_nhwc(fused[0], 256).cpu()
    assert float((got - ref).abs().max()) <= 1.6e-2 * float(ref.abs().max())


@pytest.mark.parametrize("pm,H,W,cin,cout,B", [("reflect", 128, 128, 256, 256, 2), ("zero", 12, 128, 128, 256, 3), ("reflect", 6, 192, 128, 128, 2),
                                                ("reflect", 16, 256, 128, 128, 1)],
                         ids=["resblock-512sq", "zero-pad-128w", "3-segments", "4-segments"])
def test_wgrad_row_kernel_wide_rows(pm, H, W, cin, cout, B):
    """Round 3: the image-row weight-gradient kernel on rows wider than 64 pixels (HALO variant: a K-step is one 64-pixel segment;
    the neighbouring segments' edge pixels - or the reflected pixel / zero at the image border - are staged as two extra rows).
    The ResBlock maps of the 512x512 configuration (BASELINE configs[3]) are 128 wide.  Against the generic split-K kernel on the
    same operands (2e-5 of the scale: same products, another fp32 order), against the ORACLE (bf16 tolerance), and the paired
    (two-network) launch on uneven groups against single launches."""
    u, ops, networks = _mods()
    lib, dt = u.lib.lib(), torch.bfloat16
    torch.manual_seed(H + W)
    layer = networks.ConvLayer("conv", cin, cout, 3, 1, 1, pm, dtype=dt, device="cuda")
    x = torch.rand(B, cin, H, W) * 2 - 1
    dy = torch.randn(B, cout, H, W) * 0.5
    xp, dyp = ops.to_nhwc(x.cuda(), dt), ops.to_nhwc(dy.cuda(), dt)
    assert int(lib.uig_wgrad_splits(B, H, W, cout, H, W, cin, 3, 3, 1, 1, u.lib.BF16, 512)) == max(1, min(256 // ((cout // 128) * (cin // 128) * 3), B * H)), "the image-row kernel should take this shape"
    try:
        lib.uig_debug_set_wgrad_rows(2)            # 64-wide rows only: this shape on the generic kernel
        ref = ops.conv_wgrad(layer.spec, xp, dyp)
    finally:
        lib.uig_debug_set_wgrad_rows(1)
    got = ops.conv_wgrad(layer.spec, xp, dyp)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale, f"row kernel (wide rows) vs generic: {float((got - ref).abs().max())} of {scale}"
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    xpad = F.pad(_bf(x), (1, 1, 1, 1), mode="reflect") if pm == "reflect" else F.pad(_bf(x), (1, 1, 1, 1))
    F.conv2d(xpad, w).backward(_bf(dy))
    assert (got.cpu() - w.grad).abs().max() <= _tol(dt, w.grad)
    if B >= 2:
        g = 1
        parts = ops.conv_wgrad_pair_partial(layer.spec, xp, dyp, g)
        assert parts is not None
        for sl, part in ((slice(0, g), parts[0]), (slice(g, B), parts[1])):
            a = ops.conv_wgrad(layer.spec, xp[sl], dyp[sl], partial=part)
            b = ops.conv_wgrad(layer.spec, xp[sl], dyp[sl])
            torch.cuda.synchronize()
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("B,group", [(2, 0), (4, 2), (3, 1)], ids=["single2", "paired4", "paired3-uneven"])
def test_strip_persistent_512_row_strip_on_128_wide_maps(B, group):
    """Round 3: the ResBlock FORWARD convolution of the 512x512 configuration (BASELINE configs[3]: 256 -> 256, 3x3, reflection pad 1,
    128x128 maps) on the persistent strip kernel's no-zero-row variant (256-pixel tiles = two image rows + two halo rows = 512 strip
    rows, 2 x 80 KB of LDS) instead of the generic gather kernel: output and fused InstanceNorm statistics against the oracle
    (bf16 tolerance) and against the generic kernel; kernel id asserted.  Its input gradient (zero padding: needs the zero rows)
    stays where it was."""
    u, ops, networks = _mods()
    lib, dt = u.lib.lib(), torch.bfloat16
    S, C = 128, 256
    torch.manual_seed(900 + B)
    g = group if group else B
    ls = [networks.ConvLayer("conv", C, C, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        with torch.no_grad():
            l.weight.mul_(1.5); l.bias.normal_(0, 0.1)
        l.emit_in_stats = True
        l.ensure_packed()
    x = torch.rand(B, C, S, S) * 2 - 1
    xp = ops.to_nhwc(x.cuda(), dt)
    pair = (ls[1].wp_fwd, ls[1].bias, g) if group else None
    yp = ops.conv_forward(ls[0].spec, xp, ls[0].wp_fwd, ls[0].bias, pair=pair, want_in_stats=True)
    assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK, "the 512-row strip variant did not take the launch"
    try:
        lib.uig_debug_set_strip_wide(0)
        yg = ops.conv_forward(ls[0].spec, xp, ls[0].wp_fwd, ls[0].bias, pair=pair, want_in_stats=True)
        assert lib.uig_debug_last_conv_kernel() == u.lib.K_IGEMM
    finally:
        lib.uig_debug_set_strip_wide(1)
    parts = [(0, g, 0)] + ([(g, B, 1)] if group else [])
    yref = torch.cat([F.conv2d(F.pad(_bf(x[a:e]), (1, 1, 1, 1), mode="reflect"), _bf(ls[i].weight.detach().cpu()), ls[i].bias.detach().cpu()) for a, e, i in parts])
    y = ops.from_nhwc(yp, C).cpu()
    assert (y - yref).abs().max() <= _tol(dt, yref)
    assert float((yp.float() - yg.float()).abs().max()) <= 0.02 * float(yref.abs().max())
    zn = ops.from_nhwc(networks.InstNormAct(u.lib.ACT_RELU)(yp), C).cpu()            # consumes the statistics partials of this launch
    zref = F.relu(F.instance_norm(y, eps=1e-5))
    assert (zn - zref).abs().max() <= _tol(dt, zref)


@pytest.mark.parametrize("S,cin,cout,B,group", [(64, 256, 256, 1, 0), (64, 256, 256, 3, 1), (16, 64, 128, 2, 0), (32, 128, 256, 5, 2), (24, 192, 128, 1, 0)],
                         ids=["b1-64px-infer", "b3-paired", "one-chunk", "two-chunks-paired", "three-chunks"])
def test_strip128_deep_weight_prefetch_equals_two_stage_kernel(S, cin, cout, B, group):
    """Round 3: the 128x128-tile bf16 strip kernel with FOUR weight stages (tiles fetched three K-steps ahead, counted vmcnt: an opt-in
    variant, measured slower than two stages at batch 1) must be bit-identical to the two-stage kernel (same accumulation order), forward
    with the fused InstanceNorm statistics and reflect-pad input gradient, and both close to the stock-torch convolution on
    bf16-rounded operands.  Cin = 64 / 128 / 192 / 256: 9, 18, 27, 36 K-steps (the counted wait's tail cases)."""
    u, ops, networks = _mods()
    import torch.nn.functional as F
    lib, dt = u.lib.lib(), torch.bfloat16
    torch.manual_seed(31 + S + B)
    ls = [networks.ConvLayer("conv", cin, cout, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        l.repack()
    assert lib.uig_conv_strip_tile(B, S, S, cin, cout, S, S, -1, 1, u.lib.BF16) == 128
    x = ((torch.rand(B, S, S, cin, device="cuda") * 2 - 1)).to(dt)
    dy = (torch.randn(B, S, S, cout, device="cuda") * 0.5).to(dt)
    fpair = (ls[1].wp_fwd, ls[1].bias, group) if group else None
    bpair = (ls[1].wp_dgrad, None, group) if group else None
    outs = {}
    try:
        for n in (2, 4):
            lib.uig_debug_set_strip_stages(n)
            y = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=fpair, want_in_stats=True)
            dx = ops.conv_dgrad(ls[0].spec, dy, ls[0].wp_dgrad, (S, S), bpair)
            torch.cuda.synchronize()
            outs[n] = (y.clone(), y._uig_in_partial[0].clone(), dx.clone())
    finally:
        lib.uig_debug_set_strip_stages(2)
    for a, b, name in zip(outs[2], outs[4], ("y", "InstanceNorm partial statistics", "dx")):
        assert torch.equal(a, b), f"{name}: four-stage kernel differs from the two-stage kernel"
    g = group if group else B
    xr = ops.from_nhwc(x, cin).float().cpu()
    ref = torch.cat([F.conv2d(F.pad(xr[a:e], (1, 1, 1, 1), mode="reflect"), l.weight.detach().to(dt).float().cpu(), l.bias.detach().float().cpu())
                     for (a, e), l in zip(((0, g), (g, B)), ls) if e > a])
    got = ops.from_nhwc(outs[4][0], cout).float().cpu()
    assert (got - ref).abs().max() <= 1.6e-2 * ref.abs().max()


@pytest.mark.parametrize("kind,cin,cout,B,group", [("conv", 128, 256, 3, 0), ("conv", 128, 256, 4, 2), ("convT", 256, 128, 3, 0), ("convT", 256, 128, 5, 2), ("conv", 256, 384, 2, 0)],
                         ids=["down2", "down2-pair", "up1-convT", "up1-pair-uneven", "256to384"])
def test_wgrad_row_kernel_stride2_matches_generic_and_oracle(kind, cin, cout, B, group):
    """Round 3: the stride-2 form of the image-row weight-gradient kernel (down2: Conv2d 128 -> 256 k3 s2 p1 on 128 x 128; up1:
    ConvTranspose2d 256 -> 128 k3 s2 p1 op1 on 64 x 64 - the same contraction with the maps' roles swapped): a K-step stages one
    64-pixel row of the small map and the 128-pixel row 2 i + kh - 1 of the large one, the kw taps read its pixels 2 j + kw - 1.
    Against the generic split-K kernel on the same operands (same products, different fp32 summation order: 2e-5 of max) and against
    stock torch's weight gradient on the bf16-rounded operands (bf16 products exact in fp32: 1e-4 of max), single and paired launch."""
    u, ops, networks = _mods()
    import torch.nn.functional as F
    lib, dt = u.lib.lib(), torch.bfloat16
    torch.manual_seed(90 + cin + B)
    layer = networks.ConvLayer(kind, cin, cout, 3, 2, 1, "zero", dtype=dt, device="cuda")
    spec = layer.spec
    hw = 128 if kind == "conv" else 64
    Ho, Wo = spec.out_hw(hw, hw)
    x = (torch.rand(B, hw, hw, spec.cin_p, device="cuda") * 2 - 1).to(dt)
    dy = (torch.randn(B, Ho, Wo, spec.cout_p, device="cuda") * 0.5).to(dt)
    def run():
        if group:
            parts = ops.conv_wgrad_pair_partial(spec, x, dy, group)
            assert parts is not None
            return [ops.conv_wgrad(spec, x[:group], dy[:group], partial=parts[0]), ops.conv_wgrad(spec, x[group:], dy[group:], partial=parts[1])]
        return [ops.conv_wgrad(spec, x, dy)]
    try:
        lib.uig_debug_set_wgrad_rows_s2(0)
        ref = run()
    finally:
        lib.uig_debug_set_wgrad_rows_s2(1)
    got = run()
    torch.cuda.synchronize()
    for g_, r_ in zip(got, ref):
        scale = float(r_.abs().max())
        assert scale > 0 and float((g_ - r_).abs().max()) <= 2e-5 * scale, f"stride-2 row kernel vs generic: {float((g_ - r_).abs().max())} of {scale}"
    # stock torch on the same bf16 values
    xl, dyl = ops.from_nhwc(x, spec.cin).float().cpu(), ops.from_nhwc(dy, spec.cout).float().cpu()
    w = torch.zeros(spec.weight_shape(), requires_grad=True)
    segs = ((0, group), (group, B)) if group else ((0, B),)
    for (a, e), g_ in zip(segs, got):
        y = F.conv2d(xl[a:e], w, None, 2, 1) if kind == "conv" else F.conv_transpose2d(xl[a:e], w, None, 2, 1, 1)
        (gw,) = torch.autograd.grad(y, w, dyl[a:e])
        assert float((g_.cpu() - gw).abs().max()) <= 1e-4 * float(gw.abs().max())


@pytest.mark.parametrize("S,cin,cout,B,group,pm", [(64, 256, 256, 1, 0, "reflect"), (64, 256, 256, 2, 1, "reflect"), (16, 64, 128, 3, 0, "zero"), (32, 128, 256, 5, 2, "zero"),
                                                    (20, 192, 128, 1, 0, "reflect"), (12, 128, 192, 2, 0, "zero")],
                         ids=["b1-64px-infer", "b2-paired", "one-chunk-zero", "two-chunks-paired-zero", "ragged-20px", "192-rows"])
def test_strip64_small_grid_variant_equals_128_tile_kernel(S, cin, cout, B, group, pm):
    """Round 3: on very small grids (batch-1 inference: 64 blocks of 128 x 128 on 256 CUs) plain 3x3 launches run on 64 x 64 tiles with four
    waves of 64 pixels x 16 channels (full-row stores and fused InstanceNorm statistics through the 16-channel form of the LDS-store
    epilogue).  Same K order per output element: the output must be bit-identical to the 128 x 128-tile kernel's (forward, and the plain
    zero-pad input gradient); the statistics are sums in a different order (1e-5 of their scale); ragged maps and a 192-row weight
    operand (three 64-row tiles, not a multiple of 128... refused: stays on the 128-tile kernel) included.  Both close to stock torch."""
    u, ops, networks = _mods()
    import torch.nn.functional as F
    lib, dt = u.lib.lib(), torch.bfloat16
    torch.manual_seed(43 + S + B)
    ls = [networks.ConvLayer("conv", cin, cout, 3, 1, 1, pm, dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        l.repack()
    x = ((torch.rand(B, S, S, cin, device="cuda") * 2 - 1)).to(dt)
    dy = (torch.randn(B, S, S, cout, device="cuda") * 0.5).to(dt)
    fpair = (ls[1].wp_fwd, ls[1].bias, group) if group else None
    bpair = (ls[1].wp_dgrad, None, group) if group else None
    want_stats = (S * S) % 64 == 0 and cout % 64 == 0
    outs = {}
    try:
        for mode in (2, 1):
            lib.uig_debug_set_strip_small(mode)
            y = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=fpair, want_in_stats=want_stats)
            o = [y.clone()] + ([y._uig_in_partial[0].clone()] if want_stats else [])
            if pm == "zero":
                o.append(ops.conv_dgrad(ls[0].spec, dy, ls[0].wp_dgrad, (S, S), bpair).clone())
            torch.cuda.synchronize()
            outs[mode] = o
    finally:
        lib.uig_debug_set_strip_small(2)
    # the same kernel with two weight stages instead of its default four (tiles three K-steps ahead, counted waits): bit-identical,
    # statistics included
    try:
        lib.uig_debug_set_strip_small(1); lib.uig_debug_set_strip_small_stages(2)
        y2 = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=fpair, want_in_stats=want_stats)
        torch.cuda.synchronize()
        assert torch.equal(y2, outs[1][0]) and (not want_stats or torch.equal(y2._uig_in_partial[0], outs[1][1]))
    finally:
        lib.uig_debug_set_strip_small(2); lib.uig_debug_set_strip_small_stages(4)
    # the 128 x 64-tile form (eight waves of 64 pixels x 16 channels; grids of 65 .. 128 blocks: batch-2 inference): same epilogue, same K order
    try:
        lib.uig_debug_set_strip_small(3)
        y3 = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=fpair, want_in_stats=want_stats)
        torch.cuda.synchronize()
        assert torch.equal(y3, outs[2][0])
        if want_stats:
            assert float((y3._uig_in_partial[0] - outs[2][1]).abs().max()) <= 1e-5 * float(outs[2][1].abs().max())
    finally:
        lib.uig_debug_set_strip_small(2)
    assert torch.equal(outs[2][0], outs[1][0]), "64 x 64-tile kernel output differs from the 128 x 128-tile kernel's"
    if pm == "zero":
        assert torch.equal(outs[2][-1], outs[1][-1]), "input gradient differs"
    if want_stats:
        a, b = outs[2][1], outs[1][1]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    g = group if group else B
    xr = ops.from_nhwc(x, cin).float().cpu()
    pad = (lambda t: F.pad(t, (1, 1, 1, 1), mode="reflect")) if pm == "reflect" else (lambda t: F.pad(t, (1, 1, 1, 1)))
    ref = torch.cat([F.conv2d(pad(xr[a:e]), l.weight.detach().to(dt).float().cpu(), l.bias.detach().float().cpu())
                     for (a, e), l in zip(((0, g), (g, B)), ls) if e > a])
    got = ops.from_nhwc(outs[1][0], cout).float().cpu()
    assert (got - ref).abs().max() <= 1.6e-2 * ref.abs().max()


@pytest.mark.parametrize("B,H,W,cin,cout,group,pm", [
    (16, 64, 64, 256, 256, 8, "reflect"),     # the bench launch: 512 tiles, two per block, the weight set changes between a block's tiles
    (8, 64, 64, 256, 256, 0, "reflect"),      # one tile per block
    (24, 64, 64, 256, 256, 12, "reflect"),    # three tiles per block
    (9, 64, 64, 256, 256, 4, "reflect"),      # 288 tiles: 32 blocks walk two tiles, the rest one; uneven pair
    (6, 64, 64, 256, 256, 0, "zero"),        # 192 tiles on 256 CUs: a grid smaller than the chip; zero padding (zero-row reads)
    (16, 64, 64, 128, 128, 0, "reflect"),     # two 64-channel chunks, one channel tile
    (16, 64, 32, 256, 256, 8, "zero"),       # 32-pixel-wide map: 10-line strips (320 rows)
    (12, 32, 64, 128, 256, 0, "reflect"),     # 32 x 64 map, chunk count 2, two channel tiles
    (40, 64, 64, 256, 256, 20, "reflect"),    # five tiles per block
], ids=["bench16", "single8", "three-tiles", "uneven9", "grid192-zero", "c128", "w32-zero", "h32-c128", "five-tiles"])
def test_strip_persistent_phased_schedule_equals_round3_schedule_bitwise(B, H, W, cin, cout, group, pm):
    """Round 4: the persistent strip kernel's K loop runs the PHASED schedule by default (conv_strip_pk.hip, DM 9: two wave groups one
    barrier apart, three weight stages, counted LDS-DMA waits, the third stage in what used to be spare LDS) - a new synchronisation
    structure.  Same arithmetic in the same order as round 3's loop, so every output must be BITWISE equal to it: forward + fused
    InstanceNorm partial statistics, the mirror-pixel input gradient with the skip gradient (64-wide maps) or the plain transposed
    gather.  Race screen: 12 launches per shape on fresh random data (UIG_RACE_REPS for more), over shapes that vary tiles per block (1, 2, 3, 5, mixed), grid
    size, chunk count, map width, padding and pairing.  (The oracle parity of both schedules: test_strip_persistent_256x128_bench_shape
    and the step tests.)"""
    u, ops, networks = _mods()
    lib, dt = u.lib.lib(), torch.bfloat16
    assert lib.uig_conv_strip_tile(B, H, W, cin, cout, H, W, -1, 1, u.lib.BF16) == 257
    torch.manual_seed(4000 + B + W)
    ls = [networks.ConvLayer("conv", cin, cout, 3, 1, 1, pm, dtype=dt, device="cuda") for _ in range(2)]
    for l in ls: l.repack()
    pair_f = (ls[1].wp_fwd, ls[1].bias, group) if group else None
    pair_g = (ls[1].wp_dgrad, None, group) if group else None
    gen = torch.Generator("cuda").manual_seed(17)
    try:
        for rep in range(int(os.environ.get("UIG_RACE_REPS", "12"))):      # (a one-off screen of 300 launches per shape: clean, round 4)
            x = (torch.rand(B, H, W, cin, device="cuda", generator=gen) * 2 - 1).to(dt)
            dy = (torch.randn(B, H, W, cout, device="cuda", generator=gen) * 0.5).to(dt)
            res = (torch.randn(B, H, W, cin, device="cuda", generator=gen) * 0.5).to(dt)
            out = {}
            for dm in (5, 0):
                lib.uig_debug_set_strip_pk(dm, 0)
                n0 = lib.uig_debug_strip_pk_phased_count()
                y = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=pair_f, want_in_stats=True)
                assert lib.uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK
                dx = ops.conv_dgrad(ls[0].spec, dy, ls[0].wp_dgrad, (H, W), pair=pair_g, res_add=res)
                nph = lib.uig_debug_strip_pk_phased_count() - n0      # (h32-c128: the input gradient has 96 tiles - not a persistent-kernel launch)
                assert nph == (0 if dm == 5 else (1 if (H, cin) == (32, 128) else 2)), "the launches must take the schedule under test"
                out[dm] = (y, y._uig_in_partial[0].clone(), dx)
            torch.cuda.synchronize()
            for name, a, b in zip(("y", "statistics", "dx"), out[5], out[0]):
                assert torch.equal(a, b), f"repetition {rep}: {name} of the phased schedule differs from round 3's ({int((a != b).sum())} elements)"
    finally:
        lib.uig_debug_set_strip_pk(0, 0)
