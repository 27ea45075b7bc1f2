"""Round 4: InstanceNorm statistics finalised INSIDE the launch that produces their partial slabs (arrival tickets: the image's
last-arriving block reduces the slabs in the finalize kernel's fixed order; include/uig.h, uig_conv_gather_fin) and the norm-backward
statistics carried by the mirror-pixel input-gradient launch (uig_reflect3x3_dgrad_mirror_bst).

Plus the InstanceNorm backward as ONE launch (uig_instnorm_act_bwd_fused: the blocks of an image synchronise inside the kernel).
All three are OPT-IN (UIG_IN_TICKETS=1 / UIG_MIRROR_BST=1 / UIG_FUSED_IN_BWD=1): measured on MI355X they lose to the launches they
remove (an inter-workgroup hand-off costs ~2 us per memory round trip, ~8 us per synchronisation; a kernel boundary
costs 4-5 us).  The tests switch them on.

What is pinned here: the in-launch results are BIT-IDENTICAL to the finalize launch they replace (same slabs, same fp64 association
order) - per operator at the benchmark's shapes, repeatedly and under memory load (an inter-workgroup hand-off that is wrong shows up
as stale slabs on some launches, not as a tolerance), and over whole train steps (losses and every weight, eager and graph)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods():
    import unpaired_image_generation_amd as u
    from unpaired_image_generation_amd import ops, networks
    assert u.lib.lib().uig_device_ok() == 1, "no gfx950 device visible"
    return u, ops, networks


def _finalize_ref(u, ops, y, eps=1e-5):
    """(mean, rstd) of y's fused partials through the finalize LAUNCH (round 3's path)"""
    part, nslab = y._uig_in_partial
    B, Ho, Wo, C = y.shape
    stats = torch.empty((B, C, 2), device=y.device, dtype=torch.float32)
    u.lib.check(u.lib.lib().uig_instnorm_finalize(part.data_ptr(), nslab, stats.data_ptr(), B, Ho * Wo, C, eps, torch.cuda.current_stream().cuda_stream),
                "uig_instnorm_finalize")
    return stats


CASES = [
    # kind, cin, cout, k, s, p, pad_mode, H, W, B, group        kernel family
    ("conv", 256, 256, 3, 1, 1, "reflect", 64, 64, 16, 8),    # persistent strip kernel, the benchmark's paired launch (2 tiles per block)
    ("conv", 256, 256, 3, 1, 1, "reflect", 64, 64, 8, 0),     # ... one tile per block
    ("conv", 256, 256, 3, 1, 1, "reflect", 64, 64, 12, 4),    # ... uneven groups, ragged last round
    ("conv", 256, 256, 3, 1, 1, "reflect", 64, 64, 3, 0),     # one-tile-per-block strip kernel (small grid): finalize launch behind it
    ("conv", 64, 128, 3, 2, 1, "zero", 128, 128, 8, 4),       # generic gather kernel (down1)
    ("conv", 128, 256, 3, 2, 1, "zero", 64, 64, 6, 0),        # generic gather kernel (down2)
    ("convT", 256, 128, 3, 2, 1, "zero", 32, 64, 4, 2),       # phase-fused transposed kernel (up1)
    ("conv", 128, 256, 4, 2, 1, "zero", 64, 64, 8, 4),        # PatchGAN layer 3
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}-{c[7]}x{c[8]}-b{c[9]}g{c[10]}")
def test_forward_statistics_final_out_of_the_conv_launch_bitwise(case, monkeypatch):
    kind, cin, cout, k, s, p, pm, H, W, B, group = case
    u, ops, networks = _mods()
    monkeypatch.setattr(ops, "IN_TICKETS", True)
    dt = torch.bfloat16
    torch.manual_seed(11 + B)
    ls = [networks.ConvLayer(kind, cin, cout, k, s, p, pm, dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        l.repack()
        with torch.no_grad():
            l.bias.normal_(0, 0.2)
    x = (torch.randn(B, H, W, cin, device="cuda") * 1.3 + 0.2).to(dt)
    pair = (ls[1].wp_fwd, ls[1].bias, group) if group else None
    junk = torch.empty(64 << 20, device="cuda", dtype=torch.float32)      # 256 MB: evicts L2 / MALL between the launches of the loop
    arena = ops.ticket_arena("cuda")
    with torch.no_grad():
        for it in range(12):
            if it % 3 == 1:
                junk.normal_()                                              # the next launch runs behind (and, at its head, under) a streaming kernel
            y = ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=pair, want_in_stats=True, in_eps=1e-5)
            fin = getattr(y, "_uig_in_stats", None)
            assert fin is not None and fin[1] == 1e-5
            ref = _finalize_ref(u, ops, y)
            assert torch.equal(fin[0], ref), f"iteration {it}: in-launch statistics differ from the finalize launch"
            assert int(arena.abs().sum()) == 0, "ticket words must be zero behind the launch"
    # and against the definition: mean / rstd of the stored tensor
    yf = y.float()
    mean = yf.mean(dim=(1, 2))
    var = yf.var(dim=(1, 2), unbiased=False)
    assert float((fin[0][..., 0] - mean).abs().max()) <= 1e-4 * (1 + float(mean.abs().max()))
    assert float((fin[0][..., 1] - torch.rsqrt(var + 1e-5)).abs().max()) <= 1e-3 * float(torch.rsqrt(var + 1e-5).abs().max())


@pytest.mark.parametrize("shape,act,res", [((16, 64, 64, 256), "relu", False), ((4, 128, 128, 128), "none", True), ((3, 30, 30, 512), "lrelu", False),
                                             ((8, 256, 256, 64), "relu", False)], ids=["resblock16", "res128", "patchgan-odd", "stem64"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_stand_alone_statistics_pass_finalises_itself_bitwise(shape, act, res, dtype):
    """InstanceNorm forward + backward with the norm's OWN statistics passes (no producer epilogue): ticketed (two launches each) vs the
    three-launch forms - y, saved statistics, dx and the column-sum partials bitwise equal; repeated under memory load."""
    u, ops, networks = _mods()
    if dtype == torch.float32 and shape[0] * shape[1] * shape[2] * shape[3] > (1 << 26):
        pytest.skip("large fp32 case")
    B, H, W, C = shape
    A = {"relu": u.lib.ACT_RELU, "none": u.lib.ACT_NONE, "lrelu": u.lib.ACT_LRELU}[act]
    torch.manual_seed(5)
    x = (torch.randn(shape, device="cuda") * 1.5 + 0.3).to(dtype)
    r = (torch.randn(shape, device="cuda")).to(dtype) if res else None
    dy = (torch.randn(shape, device="cuda") * 0.5).to(dtype)
    junk = torch.empty(32 << 20, device="cuda", dtype=torch.float32)

    def run(tickets):
        old = ops.IN_TICKETS
        ops.IN_TICKETS = tickets
        try:
            xr = x.clone().requires_grad_(True)
            y = ops.InstNormActFn.apply(xr, r, A, 0.2, 1e-5)
            y.backward(dy)
            return y.detach(), xr.grad, xr.grad._uig_colsum[0].clone() if hasattr(xr.grad, "_uig_colsum") else None
        finally:
            ops.IN_TICKETS = old

    y0, dx0, _ = run(False)
    for it in range(6):
        if it % 2:
            junk.normal_()
        y1, dx1, _ = run(True)
        assert torch.equal(y1, y0) and torch.equal(dx1, dx0), f"iteration {it}"
    assert int(ops.ticket_arena("cuda").abs().sum()) == 0


@pytest.mark.parametrize("B,group,act,res", [(16, 8, "relu", True), (16, 8, "none", False), (8, 0, "relu", True), (12, 4, "none", True)],
                         ids=["paired16-relu-skip", "paired16-none", "single8-relu-skip", "paired12-none-skip"])
def test_mirror_dgrad_carries_the_norm_backward_statistics(B, group, act, res, monkeypatch):
    """InstanceNorm(+act) -> 3x3 reflect conv (the ResBlock pattern, benchmark shape): the conv's mirror-pixel input-gradient launch also
    emits - final - the statistics of the norm's backward.  Against the same chain with MIRROR_BST off (the norm's own statistics
    pass): the conv's dx BITWISE equal (the variant must not change the convolution), the norm's dx within bf16 rounding (only the
    fp32 summation order of (sum g, sum g*xhat) differs: 64-pixel slabs of the epilogue vs the pass's own slabs); the statistics
    against an fp64 evaluation of their definition; in-launch finalize == finalize launch bitwise; and the oracle (autograd)."""
    u, ops, networks = _mods()
    monkeypatch.setattr(ops, "IN_TICKETS", True)
    dt = torch.bfloat16
    torch.manual_seed(300 + B)
    g = group if group else B
    ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2 if group else 1)]
    for l in ls:
        l.ensure_packed()
    A = u.lib.ACT_RELU if act == "relu" else u.lib.ACT_NONE
    x = torch.randn(B, 256, 64, 64) * 1.5 + 0.3
    dy = torch.randn(B, 256, 64, 64) * 0.5
    rs = torch.randn(B, 256, 64, 64) * 0.5

    def run(bst, lib_tickets=1):
        old = ops.MIRROR_BST
        ops.MIRROR_BST = bst
        u.lib.lib().uig_debug_set_in_tickets(lib_tickets)
        try:
            for l in ls:
                l.weight.grad = None; l.bias.grad = None
            xp = ops.to_nhwc(x.cuda(), dt).requires_grad_(True)
            h = ops.InstNormActFn.apply(xp * 1.0, None, A, 0.0, 1e-5)
            link = ops.SkipLink() if res else None
            if group:
                y = ops.PairConvFn.apply(h, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, ls[0], ls[1], g, link)
            else:
                y = ops.ConvFn.apply(h, ls[0].weight, ls[0].bias, ls[0], link)
            if res:
                link.grad = ops.to_nhwc(rs.cuda(), dt)
            hg = []
            h.register_hook(lambda t: hg.append((t.clone(), getattr(t, "_uig_bst_gm", None))))
            y.backward(ops.to_nhwc(dy.cuda(), dt))
            assert u.lib.lib().uig_debug_last_conv_kernel() == u.lib.K_STRIP_PK
            return xp.grad.clone(), hg[0][0], hg[0][1], xp.detach()
        finally:
            ops.MIRROR_BST = old
            u.lib.lib().uig_debug_set_in_tickets(1)

    dx1, dh1, gm1, xp = run(True)
    dx0, dh0, gm0, _ = run(False)
    dx2, dh2, gm2, _ = run(True, lib_tickets=0)              # the finalize LAUNCH behind the same convolution
    assert gm1 is not None and gm0 is None
    assert torch.equal(dh1, dh0), "the conv's input gradient must not change"
    assert torch.equal(gm1, gm2) and torch.equal(dx1, dx2), "in-launch finalize != finalize launch"
    sc = float(dx0.float().abs().max())
    assert float((dx1.float() - dx0.float()).abs().max()) <= 1e-2 * sc
    assert float((dx1.float() - dx0.float()).abs().mean()) <= 2e-4 * sc
    # the statistics against their definition in fp64 on the tensors the device holds
    xin = (xp * 1.0).double()                                                    # the norm's saved input (bf16 values)
    mean = xin.mean(dim=(1, 2), keepdim=True)
    rstd = torch.rsqrt(xin.var(dim=(1, 2), unbiased=False, keepdim=True) + 1e-5)
    xh = (xin - mean) * rstd
    gg = dh1.double() * ((xh > 0).double() if act == "relu" else 1.0)
    ref = torch.stack([gg.mean(dim=(1, 2)), (gg * xh).mean(dim=(1, 2))], dim=-1)
    err = float((gm1.double() - ref).abs().max())
    assert err <= 2e-3 * float(ref.abs().max()) + 1e-6, err                      # (mean, rstd) themselves are fp32 on the device
    # oracle
    xr = x.to(torch.bfloat16).float().requires_grad_(True)
    hr = F.instance_norm(xr, eps=1e-5)
    hr = F.relu(hr) if act == "relu" else hr
    hb = hr.detach().to(torch.bfloat16).float().requires_grad_(True)
    bf = lambda t: t.to(torch.bfloat16).float()
    parts = [(0, g, 0)] + ([(g, B, 1)] if group else [])
    yr = torch.cat([F.conv2d(F.pad(hb[a:e], (1, 1, 1, 1), mode="reflect"), bf(ls[i].weight.detach().cpu()), ls[i].bias.detach().cpu()) for a, e, i in parts])
    yr.backward(bf(dy))
    hr.backward(bf(hb.grad + (bf(rs) if res else 0)))
    got = ops.from_nhwc(dx1, 256).cpu()
    assert (got - xr.grad).abs().max() <= 2.5e-2 * float(xr.grad.abs().max())


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_train_step_with_in_launch_finalize_bitwise_equals_finalize_launches(use_graph):
    """configs[1]'s step (9 blocks, B = 4, 256x256, bf16), three steps: arrival-ticket finalize (opt-in) vs the finalize launches
    (ops.IN_TICKETS = False: round 3's kernel sequence) - the 8 losses of every step and all weights afterwards bitwise equal."""
    u, ops, networks = _mods()
    res = {}
    for tickets in (True, False):
        old = ops.IN_TICKETS
        ops.IN_TICKETS = tickets
        try:
            torch.manual_seed(0)
            m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, device="cuda", use_graph=use_graph)
            torch.manual_seed(1)
            rA = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
            rB = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
            ls = [m.train_step(rA, rB, sync=False).clone() for _ in range(3)]
            torch.cuda.synchronize()
            res[tickets] = (torch.stack(ls).cpu(), m.grp_G.flat.clone().cpu(), m.grp_D.flat.clone().cpu())
            m.close()
            del m
        finally:
            ops.IN_TICKETS = old
    for a, b, name in zip(res[True], res[False], ("losses", "generator weights", "discriminator weights")):
        assert torch.equal(a, b), f"{name} differ between the in-launch finalize and the finalize launches"
    assert bool(torch.isfinite(res[True][0]).all())


def test_train_step_mirror_bst_tracks_the_separate_statistics_pass(monkeypatch):
    """MIRROR_BST on (default) vs off over two steps of configs[1]'s workload: not bitwise (the fp32 summation order of the norm-backward
    statistics differs) - the losses agree to 2e-3 relative after one update, the generator weights to 1e-4 of their scale."""
    u, ops, networks = _mods()
    monkeypatch.setattr(ops, "IN_TICKETS", True)
    res = {}
    for bst in (True, False):
        old = ops.MIRROR_BST
        ops.MIRROR_BST = bst
        try:
            torch.manual_seed(0)
            m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, device="cuda", use_graph=False)
            torch.manual_seed(1)
            rA = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
            rB = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
            ls = [m.train_step(rA, rB, sync=False).clone() for _ in range(2)]
            torch.cuda.synchronize()
            res[bst] = (torch.stack(ls).cpu(), m.grp_G.flat.clone().cpu())
            m.close()
            del m
        finally:
            ops.MIRROR_BST = old
    l1, l0 = res[True][0], res[False][0]
    assert torch.equal(l1[0, :6], l0[0, :6]) or float(((l1[0] - l0[0]).abs() / l0[0].abs()).max()) <= 1e-5      # step 0's forward is the same computation
    assert float(((l1[1] - l0[1]).abs() / l0[1].abs()).max()) <= 2e-3, (l1[1], l0[1])
    w1, w0 = res[True][1], res[False][1]
    # Adam's first steps move every weight by ~lr = 2e-4 whatever the gradient's size: a gradient near zero that changes sign moves its
    # weight by up to 2 lr per step; on average the two runs stay together
    assert float((w1 - w0).abs().max()) <= 1e-3 and float((w1 - w0).abs().mean()) <= 1e-4, (float((w1 - w0).abs().max()), float((w1 - w0).abs().mean()))


@pytest.mark.parametrize("shape,act,dtype", [((16, 64, 64, 256), "relu", torch.bfloat16), ((8, 64, 64, 256), "none", torch.bfloat16), ((12, 64, 64, 256), "relu", torch.bfloat16),
                                             ((16, 64, 64, 128), "lrelu", torch.bfloat16), ((16, 32, 32, 256), "lrelu", torch.bfloat16), ((5, 31, 31, 512), "lrelu", torch.bfloat16),
                                             ((3, 24, 40, 64), "relu", torch.bfloat16), ((2, 32, 32, 128), "relu", torch.float32)],
                         ids=["resblock16", "resblock8-none", "resblock12", "patchgan-128", "patchgan-256", "patchgan-512-odd", "small-64ch", "f32"])
def test_fused_instnorm_backward_one_launch_vs_three_launch_form_and_oracle(shape, act, dtype, monkeypatch):
    """Round 4: InstanceNorm backward as ONE launch and one pass over dy and x (uig_instnorm_act_bwd_fused: statistics, finalize and
    apply fused; the blocks of an image synchronise inside the kernel and keep their pixels in registers across the waits).
    Against the three-launch form: same arithmetic, the statistics' fp32 partial sums cover other pixel ranges - dx within bf16 rounding
    (max 1e-2 of max, mean 2e-4), the bias gradient from the column-sum partials to 1e-5 of sum |dx|; against the oracle
    (F.instance_norm autograd on the same bf16 inputs), 1.6e-2 of max; run-to-run BITWISE reproducible under memory load (the in-kernel
    hand-offs: a stale read would show as a different result on some run); the kernel's error word stays 0."""
    u, ops, networks = _mods()
    B, H, W, C = shape
    lib = u.lib.lib()
    dt = u.lib.BF16 if dtype == torch.bfloat16 else u.lib.F32
    assert lib.uig_instnorm_bwd_fused_applicable(B, H * W, C, dt) > 0
    A = {"relu": u.lib.ACT_RELU, "none": u.lib.ACT_NONE, "lrelu": u.lib.ACT_LRELU}[act]
    torch.manual_seed(9)
    x = (torch.randn(shape, device="cuda") * 1.5 + 0.3).to(dtype)
    dy = (torch.randn(shape, device="cuda") * 0.5).to(dtype)
    junk = torch.empty(32 << 20, device="cuda", dtype=torch.float32)

    def run(fused):
        monkeypatch.setattr(ops, "FUSED_IN_BWD", fused)
        xr = x.clone().requires_grad_(True)
        xin = xr * 1.0                                      # non-leaf: its gradient tensor (with the column-sum attribute) reaches the hook
        got = []
        xin.register_hook(lambda t: got.append(t))
        y = ops.InstNormActFn.apply(xin, None, A, 0.2, 1e-5)
        y.backward(dy)
        g = got[0]
        db = ops._bias_grad_from_partials(g._uig_colsum, 0, B, C, None, False)
        return g.detach().clone(), db

    dx0, db0 = run(False)
    outs = []
    for it in range(5):
        if it % 2:
            junk.normal_()
        outs.append(run(True))
    dx1, db1 = outs[0]
    for k, (a, b) in enumerate(outs[1:]):
        assert torch.equal(a, dx1) and torch.equal(b, db1), f"run {k + 1} differs from run 0"
    ops.check_sync_errors("cuda")
    sc = float(dx0.float().abs().max())
    d = (dx1.float() - dx0.float()).abs()
    tol_max, tol_mean = (1e-2, 2e-4) if dtype == torch.bfloat16 else (2e-5, 2e-6)
    assert float(d.max()) <= tol_max * sc and float(d.mean()) <= tol_mean * sc, (float(d.max()) / sc, float(d.mean()) / sc)
    assert float((db1 - db0).abs().max()) <= 1e-5 * float(dx0.float().abs().sum((0, 1, 2)).max()) + 1e-6
    # oracle
    xc = x.float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yr = F.instance_norm(xc, eps=1e-5)
    yr = F.relu(yr) if act == "relu" else (F.leaky_relu(yr, 0.2) if act == "lrelu" else yr)
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = xc.grad.permute(0, 2, 3, 1)
    assert float((dx1.float().cpu() - ref).abs().max()) <= (1.6e-2 if dtype == torch.bfloat16 else 2e-5) * float(ref.abs().max()) + 1e-6
    assert int(ops.ticket_arena("cuda").abs().sum()) == 0, "synchronisation words must be zero behind the launches"


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_train_step_fused_instnorm_backward_tracks_the_three_launch_form(use_graph, monkeypatch):
    """configs[1]'s step (9 blocks, B = 4, 256x256, bf16), two steps with the fused InstanceNorm backward (opt-in) vs the three-launch
    form: step 0's losses are the same forward computation (bitwise), step 1's agree to 2e-3 relative, the weights stay together."""
    u, ops, networks = _mods()
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_IN_BWD", fused)
        torch.manual_seed(0)
        m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, device="cuda", use_graph=use_graph)
        torch.manual_seed(1)
        rA = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
        rB = torch.rand(4, 3, 256, 256, device="cuda") * 2 - 1
        ls = [m.train_step(rA, rB, sync=False).clone() for _ in range(2)]
        torch.cuda.synchronize()
        ops.check_sync_errors("cuda")
        res[fused] = (torch.stack(ls).cpu(), m.grp_G.flat.clone().cpu())
        m.close()
        del m
    l1, l0 = res[True][0], res[False][0]
    assert torch.equal(l1[0, :6], l0[0, :6])
    assert float(((l1[1] - l0[1]).abs() / l0[1].abs()).max()) <= 2e-3, (l1[1], l0[1])
    w1, w0 = res[True][1], res[False][1]
    assert float((w1 - w0).abs().max()) <= 1e-3 and float((w1 - w0).abs().mean()) <= 1e-4
