"""GPU parity of the modules and the train step against the committed golden fixtures (tests/golden, generated from
the CPU oracle by tests/golden/make_golden.py) and against the oracle run in-process on the same seeded inputs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_oracle_weights(model, o):
    model.load_state_dicts(o.G_A.state_dict(), o.G_B.state_dict(), o.D_A.state_dict(), o.D_B.state_dict())


def test_generator_config1_fp32_linf():
    """BASELINE.json configs[0] + north_star gate: generator output L-inf < 1e-3 vs the CPU reference (fp32 path)."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Generator as OG, init_weights
    gold = np.load(os.path.join(GOLD, "config1_g6_64.npz"))
    torch.manual_seed(1234)
    og = init_weights(OG(n_blocks=6))
    g = u.Generator(n_blocks=6, dtype=torch.float32)
    g.load_state_dict(og.state_dict())
    x = torch.from_numpy(gold["x"])
    with torch.no_grad():
        y = g(x.cuda()).cpu()
    linf = float((y - torch.from_numpy(gold["y"])).abs().max())
    print("config1 fp32 L-inf vs golden:", linf)
    assert y.shape == (1, 3, 64, 64) and linf < 1e-3
    # bf16 path: stated tolerance 0.12 absolute on the tanh output (bf16 drifts 4-7e-2 from fp32)
    gb = u.Generator(n_blocks=6, dtype=torch.bfloat16)
    gb.load_state_dict(og.state_dict())
    with torch.no_grad():
        yb = gb(x.cuda()).cpu()
    linf_b = float((yb - torch.from_numpy(gold["y"])).abs().max())
    print("config1 bf16 L-inf vs golden:", linf_b)
    assert linf_b < 0.12


def test_discriminator_golden():
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Discriminator as OD, init_weights
    gold = np.load(os.path.join(GOLD, "disc_64.npz"))
    torch.manual_seed(4321)
    od = init_weights(OD())
    d = u.Discriminator(dtype=torch.float32)
    d.load_state_dict(od.state_dict())
    with torch.no_grad():
        y = d(torch.from_numpy(gold["x"]).cuda()).cpu()
    assert y.shape == gold["y"].shape
    assert float((y - torch.from_numpy(gold["y"])).abs().max()) < 1e-3


def test_generator_256_9block_fp32_linf():
    """The headline parity gate at the benchmark's own size: G(9) @ 256x256, L-inf < 1e-3 vs the oracle run here."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Generator as OG, init_weights
    torch.manual_seed(2)
    og = init_weights(OG(n_blocks=9))
    x = torch.rand(1, 3, 256, 256) * 2 - 1
    with torch.no_grad():
        yref = og(x)
    g = u.Generator(n_blocks=9, dtype=torch.float32)
    g.load_state_dict(og.state_dict())
    with torch.no_grad():
        y = g(x.cuda()).cpu()
    linf = float((y - yref).abs().max())
    print("G9@256 fp32 L-inf:", linf)
    assert linf < 1e-3


def _golden_step_check(model, rel):
    gold = np.load(os.path.join(GOLD, "train_step_64.npz"))
    want = json.load(open(os.path.join(GOLD, "train_step_64_losses.json")))
    rA, rB = torch.from_numpy(gold["real_A"]).cuda(), torch.from_numpy(gold["real_B"]).cuda()
    for step in range(2):
        got = model.train_step(rA, rB)
        # step 1 runs on weights that already took one Adam step (update = lr * m/sqrt(v): sign-like, so fp32 rounding
        # differences in tiny gradients are amplified) -> 10x the step-0 tolerance there
        tol = rel * (1 if step == 0 else 10)
        for k, v in want[step].items():
            assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (step, k, got[k], v)
    return gold


@pytest.mark.parametrize("fused", ["paired", "batched", "unbatched"])
def test_train_step_golden_fp32(fused):
    """Two full §3.1 steps (B=2, 64x64, G6) on the fp32 path: 8 losses per step + generated image vs golden."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import CycleGANOracle
    torch.manual_seed(7)
    o = CycleGANOracle(n_blocks=6)
    m = u.CycleGAN(n_blocks=6, dtype=torch.float32, batch_fused=fused != "unbatched", paired=fused == "paired")
    _load_oracle_weights(m, o)
    gold = _golden_step_check(m, 2e-4)
    fb = u.ops.from_nhwc(m.last_fake_B, 3).cpu() if hasattr(m, "last_fake_B") else None
    if fb is not None:
        # generated image of step 2, i.e. after one Adam step (+-lr per element, sign decided by noise-sized gradients
        # for some elements): a few 1e-2 on the tanh output; step-0 outputs are pinned to 1e-3 by the tests above
        assert float((fb - torch.from_numpy(gold["fake_B"])).abs().max()) < 5e-2


def test_train_step_bf16_tracks_oracle():
    """bf16 compute path: losses within 3 % of the fp32 oracle on the same inputs (stated bf16 tolerance)."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import CycleGANOracle
    torch.manual_seed(7)
    o = CycleGANOracle(n_blocks=6)
    m = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16)
    _load_oracle_weights(m, o)
    _golden_step_check(m, 3e-2)


def test_graph_replay_equals_eager():
    """HIP-graph replay of the step is bitwise the eager step (same kernels, same order)."""
    import unpaired_image_generation_amd as u
    torch.manual_seed(21)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    torch.manual_seed(5)
    me = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=False)
    mg = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=True)
    mg.load_state_dicts(*[n.state_dict() for n in me.nets()])
    for step in range(3):
        le = me.train_step(rA, rB); lg = mg.train_step(rA, rB)
        assert le == lg, (step, le, lg)
    assert torch.equal(me.grp_G.flat, mg.grp_G.flat) and torch.equal(me.grp_D.flat, mg.grp_D.flat)


def test_schedule_pool_and_checkpoint_graph_equals_eager(tmp_path):
    """§8(f) rows 1-2 on the device: the LR-schedule multiplier reaches the graph-replayed Adam through its device record,
    the image pools feed the discriminator graph through static buffers, and a checkpoint resumes bit-identically."""
    import unpaired_image_generation_amd as u
    torch.manual_seed(31)
    batches = [tuple(torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2)) for _ in range(5)]
    torch.manual_seed(6)
    me = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=False, pool_size=3, pool_seed=4)
    mg = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=True, pool_size=3, pool_seed=4)
    mg.load_state_dicts(*[n.state_dict() for n in me.nets()])
    w0 = me.grp_G.flat.clone()
    for step, (rA, rB) in enumerate(batches[:3]):
        if step == 2:
            me.set_epoch(150, 100, 100); mg.set_epoch(150, 100, 100)      # LR x (1 - 50/101)
        le = me.train_step(rA, rB); lg = mg.train_step(rA, rB)
        assert le == lg, (step, le, lg)
    assert torch.equal(me.grp_G.flat, mg.grp_G.flat) and torch.equal(me.grp_D.flat, mg.grp_D.flat)
    assert not torch.equal(me.grp_G.flat, w0)
    assert me.pool_B.n == 3 and torch.equal(me.pool_B.buf, mg.pool_B.buf)
    # checkpoint from the graph model -> a fresh eager model resumes exactly where both are
    path = str(tmp_path / "ckpt.pt")
    mg.save(path)
    torch.manual_seed(123)
    mr = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=False, pool_size=3, pool_seed=77)
    mr.load(path)
    assert mr.lr_scale == me.lr_scale and mr.grp_G.step == 3
    for rA, rB in batches[3:]:
        le = me.train_step(rA, rB); lr_ = mr.train_step(rA, rB); lg = mg.train_step(rA, rB)
        assert le == lr_ == lg
    assert torch.equal(me.grp_G.flat, mr.grp_G.flat) and torch.equal(me.grp_D.m, mr.grp_D.m)


def test_lr_scale_scales_the_update():
    """Adam's first step moves every weight by lr * scale (bias-corrected m / sqrt(v) = +-1): halving the multiplier halves it."""
    import unpaired_image_generation_amd as u
    torch.manual_seed(8)
    rA, rB = (torch.rand(1, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    deltas = []
    for scale, graph in ((1.0, False), (0.5, False), (0.5, True)):
        torch.manual_seed(9)
        m = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=graph)
        w0 = m.grp_D.flat.clone()
        m.set_lr_scale(scale)
        m.train_step(rA, rB)
        deltas.append((m.grp_D.flat - w0).abs().max().item())
    assert abs(deltas[0] - 2e-4) < 2e-6 and abs(deltas[1] - 1e-4) < 1e-6 and abs(deltas[2] - 1e-4) < 1e-6, deltas


def test_train_step_gradients_vs_oracle_fp32():
    """The whole backward path at once: after one §3.1 step, every parameter gradient (G_A, G_B, D_A, D_B) vs the
    oracle's autograd on the same weights and inputs.  fp32 path; relative L2 error per tensor < 1e-2 (measured worst ~3e-3, on the 7x7 stem whose gradient crosses the whole net).
    Biases that feed an InstanceNorm have a mathematically zero gradient (rounding noise in both implementations):
    for them only the magnitude is checked.  Then the post-Adam weights: the first Adam step moves every element by
    exactly +-lr, so agreement means the SIGN of each gradient element agrees wherever it is not noise-sized."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import CycleGANOracle
    torch.manual_seed(11)
    o = CycleGANOracle(n_blocks=6)
    m = u.CycleGAN(n_blocks=6, dtype=torch.float32)
    _load_oracle_weights(m, o)
    rA, rB = torch.rand(2, 3, 64, 64) * 2 - 1, torch.rand(2, 3, 64, 64) * 2 - 1
    w0 = {id(n): {k: v.clone() for k, v in n.state_dict().items()} for n in o.nets()}
    lo = o.train_step(rA, rB)
    lm = m.train_step(rA.cuda(), rB.cuda())
    for k in lo:
        assert abs(lo[k] - lm[k]) < 2e-4 * max(1.0, abs(lo[k])), (k, lo[k], lm[k])
    worst = 0.0
    for name, on, mn in (("G_A", o.G_A, m.G_A), ("G_B", o.G_B, m.G_B), ("D_A", o.D_A, m.D_A), ("D_B", o.D_B, m.D_B)):
        mp = dict(mn.named_parameters())
        n_in_front_of_norm = 0
        for k, p in on.named_parameters():
            g, gr = mp[k].grad.cpu(), p.grad
            ref = float(gr.norm())
            wscale = float(dict(on.named_parameters())[k.replace(".bias", ".weight")].grad.norm())
            if k.endswith(".bias") and ref < 1e-4 * wscale:      # bias feeding an InstanceNorm: zero gradient + noise
                assert float(g.norm()) < 1e-3 * wscale, (name, k, float(g.norm()), wscale)
                n_in_front_of_norm += 1
                continue
            rel = float((g - gr).norm()) / (ref + 1e-30)
            worst = max(worst, rel)
            assert rel < 1e-2, (name, k, rel, ref)
        assert n_in_front_of_norm == (len(list(on.parameters())) // 2 - 1 if name[0] == "G" else 3), (name, n_in_front_of_norm)
    print("worst relative L2 gradient error:", worst)
    # post-step weights: |delta| == lr per element after Adam step 1; compare where the oracle gradient is not noise
    for on, mn in ((o.G_A, m.G_A), (o.D_B, m.D_B)):
        mp = dict(mn.named_parameters())
        for k, p in on.named_parameters():
            if not k.endswith(".weight"):
                continue
            gr = p.grad
            big = gr.abs() > 0.05 * gr.abs().max()      # well above the ~6e-3 relative gradient error
            d = (mp[k].detach().cpu() - p.detach()).abs()
            assert float(d[big].max()) < 2e-5, (k, float(d[big].max()))
            moved = (p.detach() - w0[id(on)][k]).abs()
            assert float((moved[big] - 2e-4).abs().max()) < 2e-5


@pytest.mark.parametrize("B,H,W", [(1, 32, 32), (2, 48, 48), (1, 72, 104), (3, 128, 128), (1, 96, 64)])
def test_generator_shapes_fwd_bwd_fp32(B, H, W):
    """Shape sweep (tiny / non-square / odd tile counts / batch 1 and 3): generator forward L-inf < 1e-3 and the input
    gradient + every weight gradient vs the oracle.  Exercises the strip / generic / border / fold fallbacks.
    Gradient tolerance: relative L2 < 1e-2.  Per operator the HIP kernels agree with the oracle to ~1e-6 (tests/test_ops_gpu.py, fp32 cases);
    through the network a handful of pre-activations lie within rounding distance of 0, the ReLU mask of those elements
    flips between two correct fp32 evaluations and everything below inherits an O(1e-3) relative difference - the fp32
    oracle differs from the fp64 oracle by 8e-4 / 1.2e-3 on the 64x64 / 128x128 cases here, and by 1e-6 at 32x32."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Generator as OG, init_weights
    torch.manual_seed(100 + H + W)
    og = init_weights(OG(n_blocks=2))
    g = u.Generator(n_blocks=2, dtype=torch.float32)
    g.load_state_dict(og.state_dict())
    x = torch.rand(B, 3, H, W) * 2 - 1
    xr = x.clone().requires_grad_(True)
    yr = og(xr)
    t = torch.randn_like(yr)
    (yr * t).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = g(xg)
    assert float((y.detach().cpu() - yr.detach()).abs().max()) < 1e-3
    (y * t.cuda()).sum().backward()
    assert float((xg.grad.cpu() - xr.grad).norm() / xr.grad.norm()) < 1e-2
    ref = dict(og.named_parameters())
    for k, p in g.named_parameters():
        if k.endswith(".weight"):
            r = ref[k].grad
            assert float((p.grad.cpu() - r).norm() / (r.norm() + 1e-30)) < 1e-2, k


def test_generator_512_fp32_linf_and_transposed_layers():
    """BASELINE.json configs[3] (512x512 9-block generator, ConvTranspose2d stress): fp32 forward L-inf < 1e-3 vs the oracle,
    and the bf16 forward (phase-fused transposed kernel on the 128-wide up-sampling layer, generic kernel on the 256-wide
    one) within the bf16 tolerance of the 256x256 test."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Generator as OG, init_weights
    torch.manual_seed(11)
    og = init_weights(OG(n_blocks=9))
    x = torch.rand(1, 3, 512, 512) * 2 - 1
    with torch.no_grad():
        yref = og(x)
    g = u.Generator(n_blocks=9, dtype=torch.float32)
    g.load_state_dict(og.state_dict())
    with torch.no_grad():
        y = g(x.cuda()).cpu()
    linf = float((y - yref).abs().max())
    print("G9@512 fp32 L-inf:", linf)
    assert y.shape == (1, 3, 512, 512) and linf < 1e-3
    gb = u.Generator(n_blocks=9, dtype=torch.bfloat16)
    gb.load_state_dict(og.state_dict())
    with torch.no_grad():
        yb = gb(x.cuda()).cpu()
    linf_b = float((yb - yref).abs().max())
    print("G9@512 bf16 L-inf:", linf_b)
    assert linf_b < 0.12


def test_train_step_512_batch2_graph_equals_eager():
    """configs[3] as a train step: batch 2 at 512x512 (bf16).  Graph replay and eager launches run the same kernels: the
    8 losses agree exactly over 2 steps and stay finite."""
    import unpaired_image_generation_amd as u
    torch.manual_seed(21)
    a = torch.rand(2, 3, 512, 512, device="cuda") * 2 - 1
    b = torch.rand(2, 3, 512, 512, device="cuda") * 2 - 1
    torch.manual_seed(5)
    m1 = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=False)
    m2 = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=True)
    m2.load_state_dicts(*[n.state_dict() for n in m1.nets()])
    for _ in range(2):
        l1, l2 = m1.train_step(a, b), m2.train_step(a, b)
        for k in l1:
            assert np.isfinite(l1[k]) and l1[k] == l2[k], (k, l1[k], l2[k])


def test_train_step_config2_b4_256_bf16_graph_vs_oracle():
    """BASELINE.json configs[1], the workload bench.py measures: 9-block G_A/G_B + PatchGAN D_A/D_B, batch 4 at 256x256, bf16
    MFMA path, paired launches, HIP-graph replay.  The 8 losses of the first step against the fp32 CPU oracle on the same
    weights and inputs, stated bf16 tolerance 3 % (bf16 operands drift 4-7e-2 on the generator output; the
    losses are means over >= 3600 elements); second step (weights after one Adam step of +-lr per element) 10 %.
    Round 4: the oracle steps (fp32 steps 0 and 1, bf16 same-rounding emulation step 0) are the committed fixture
    tests/golden/step_config2_b4_256_bf16.npz (tests/golden/make_step_golden.py; weights and inputs regenerated here from the seed and
    checked against its checksums); tensors are compared on the fixture's strided 64 K-element samples.  What pins every kernel of
    this step element-wise is tests/test_teacher_forced_gpu.py."""
    import unpaired_image_generation_amd as u
    import _step_golden as SG
    from oracle.torch_oracle import CycleGANOracle
    gold = SG.load("step_config2_b4_256_bf16.npz")
    torch.manual_seed(3)
    o = CycleGANOracle(n_blocks=9)                      # weights only: no oracle step runs here
    m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=True)
    _load_oracle_weights(m, o)
    rA, rB = torch.rand(4, 3, 256, 256) * 2 - 1, torch.rand(4, 3, 256, 256) * 2 - 1
    SG.assert_same_problem(gold, o, rA, rB)
    for step, tol in ((0, 3e-2), (1, 1e-1)):
        lo = SG.losses(gold, f"loss_fp32_step{step}")
        lm = m.train_step(rA.cuda(), rB.cuda())
        assert m.graph_active, "the step fell back to eager launches"
        print(f"step {step}:", {k: (round(lo[k], 4), round(lm[k], 4)) for k in lo})
        for k in lo:
            assert abs(lo[k] - lm[k]) <= tol * max(1.0, abs(lo[k])), (step, k, lo[k], lm[k])
        if step == 0:
            le = SG.losses(gold, "loss_emu_step0")
            print("step 0 vs bf16 emulation:", {k: (round(le[k], 4), round(lm[k], 4)) for k in le})
            for k in le:      # stated: 1 % (same roundings; what is left is fp32 summation order across rounding boundaries)
                assert abs(le[k] - lm[k]) <= 1e-2 * max(1.0, abs(le[k])), (k, le[k], lm[k])
            fb = SG.sample(u.ops.from_nhwc(m.last_fake_B, 3))
            d = (fb - torch.from_numpy(gold["fake_B_emu"])).abs()
            d32 = (fb - torch.from_numpy(gold["fake_B_fp32"])).abs()
            print("fake_B vs bf16 emulation (64 K-element sample): L-inf", float(d.max()), "mean", float(d.mean()), "| vs fp32 oracle L-inf", float(d32.max()))
            assert float(d.max()) <= 0.12 and float(d.mean()) <= 1e-2      # measured 0.049 / 0.0056
            assert float(d32.max()) <= 0.12                                 # bf16 drifts 4-7e-2 from fp32 on the tanh output
            for name, mine, key in (("G_A ResBlock 5 conv 2", m.G_A[14].b[5].weight.grad, "grad_emu_G_A.14.b.5"),
                                    ("G_B up1", m.G_B[19].weight.grad, "grad_emu_G_B.19"),
                                    ("D_A 256->512", m.D_A[8].weight.grad, "grad_emu_D_A.8")):
                rel, cos = SG.rel_cos(mine, gold[key])
                print(f"weight gradient {name}: relative L2 vs bf16 emulation {rel:.3e}, cosine {cos:.4f}")
                # Stated: relative L2 <= 0.40, cosine >= 0.93 [measured 0.256 / 0.967 on the ResBlock conv].  The same roundings do not
                # make the two runs agree element-wise through 30 layers: one conv output in ~2000 lands on the other side of a bf16
                # rounding boundary because the fp32 summation ORDER differs, the next layers spread that over every output, and
                # after ~4 layers the trajectories carry independent bf16 noise; ReLU masks of near-zero pre-activations then
                # differ and re-route those elements' gradients - the emulation itself sits 24 % from the fp32 oracle on such a
                # gradient (tests/test_lowprec_oracle.py).  A wrong or missing term is uncorrelated (cosine ~0) and fails this;
                # exact per-kernel agreement of THIS step, layer by layer, is tests/test_teacher_forced_gpu.py.
                assert rel <= 0.40 and cos >= 0.93, (name, rel, cos)
    m.close()


def test_train_step_config4_b2_512_bf16_graph_vs_same_rounding_oracle():
    """BASELINE.json configs[3] AT ITS OWN WORKLOAD (round 3): batch 2 at 512x512, 9-block generators, bf16, HIP-graph replay - the
    configuration whose 128-pixel-wide ResBlock maps run on the wide-row weight-gradient kernel and the 512-row forward strip.
    One full train step against the same-rounding CPU emulation of the bf16 step (oracle/lowprec_oracle.LowPrecOracle; round 4: its
    results are the committed fixture tests/golden/step_config4_b2_512_bf16.npz): 8 losses to 1 % [measured <= 1e-3], generated image
    mean |diff| <= 1e-2 / L-inf <= 0.12, one ResBlock weight gradient relative L2 <= 0.40 and cosine >= 0.93 (why not tighter: see
    test_train_step_config2_b4_256_bf16_graph_vs_oracle).  PARITY UNPINNED BY THE REFERENCE."""
    import unpaired_image_generation_amd as u
    import _step_golden as SG
    from oracle.torch_oracle import CycleGANOracle
    gold = SG.load("step_config4_b2_512_bf16.npz")
    torch.manual_seed(13)
    e = CycleGANOracle(n_blocks=9)                      # the emulation's weights (same constructor, same seed); no oracle step runs here
    m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, use_graph=True)
    _load_oracle_weights(m, e)
    rA, rB = torch.rand(2, 3, 512, 512) * 2 - 1, torch.rand(2, 3, 512, 512) * 2 - 1
    SG.assert_same_problem(gold, e, rA, rB)
    lm = m.train_step(rA.cuda(), rB.cuda())
    assert m.graph_active, "the step fell back to eager launches"
    le = SG.losses(gold, "loss_emu_step0")
    print({k: (round(le[k], 4), round(lm[k], 4)) for k in le})
    for k in le:
        assert lm[k] == lm[k] and abs(le[k] - lm[k]) <= 1e-2 * max(1.0, abs(le[k])), (k, le[k], lm[k])
    d = (SG.sample(u.ops.from_nhwc(m.last_fake_B, 3)) - torch.from_numpy(gold["fake_B_emu"])).abs()
    print("fake_B vs bf16 emulation (64 K-element sample): L-inf", float(d.max()), "mean", float(d.mean()))
    assert float(d.max()) <= 0.12 and float(d.mean()) <= 1e-2
    rel, cos = SG.rel_cos(m.G_A[14].b[5].weight.grad, gold["grad_emu_G_A.14.b.5"])
    print(f"weight gradient G_A ResBlock 5 conv 2: relative L2 vs bf16 emulation {rel:.3e}, cosine {cos:.4f}")
    assert rel <= 0.40 and cos >= 0.93, (rel, cos)
    m.close()


def test_train_step_256_fp32_vs_committed_golden():
    """SURVEY Appendix B recipe B2 (seed 0, B=1, 256x256, 9 blocks): the 8 first-step losses committed in
    tests/golden/train_step_256_losses.json, reproduced by the exact-f32 HIP path to 2e-4 relative.  Weights and inputs are
    regenerated from the seed by constructing the oracle (no oracle step is run here)."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import CycleGANOracle
    gold = json.load(open(os.path.join(GOLD, "train_step_256_losses.json")))
    torch.manual_seed(gold["seed"])
    o = CycleGANOracle(n_blocks=9)
    rA = torch.rand(gold["B"], 3, gold["H"], gold["H"]) * 2 - 1
    rB = torch.rand(gold["B"], 3, gold["H"], gold["H"]) * 2 - 1
    m = u.CycleGAN(n_blocks=9, dtype=torch.float32)
    _load_oracle_weights(m, o)
    lm = m.train_step(rA.cuda(), rB.cuda())
    print({k: (v, lm[k]) for k, v in gold["losses"].items()})
    for k, v in gold["losses"].items():
        assert abs(lm[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, lm[k], v)


def test_pool_schedule_resume_vs_oracle_fp32(tmp_path):
    """SURVEY §8(f) rows 1-2 against the ORACLE (not self-comparison): 4 steps with 3-image history pools (same seeds, same
    host RNG draw order), the recipe's LR decay switched on before step 2, a checkpoint after step 2 resumed in a fresh
    model; fp32 path, graph replay for the first model, eager for the resumed one.  Losses per step vs the oracle's
    (stock LambdaLR on stock Adam, list-based pool): 2e-4 relative at step 0, 2e-3 afterwards (Adam's sign-like first
    updates amplify fp32 rounding differences of noise-sized gradients)."""
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import CycleGANOracle
    torch.manual_seed(17)
    o = CycleGANOracle(n_blocks=6, pool_size=3, pool_seed=9)
    batches = [(torch.rand(2, 3, 64, 64) * 2 - 1, torch.rand(2, 3, 64, 64) * 2 - 1) for _ in range(4)]
    m = u.CycleGAN(n_blocks=6, dtype=torch.float32, use_graph=True, pool_size=3, pool_seed=9)
    _load_oracle_weights(m, o)

    def check(step, lo, lm, what):
        tol = (2e-4, 2e-3, 2e-2, 4e-2)[step]
        for k in lo:
            assert abs(lo[k] - lm[k]) <= tol * max(1.0, abs(lo[k])), (what, step, k, lo[k], lm[k])

    mr = None
    upd = {}
    for step, (rA, rB) in enumerate(batches):
        w_hip, w_or = m.grp_D.flat.clone(), torch.cat([p.detach().reshape(-1) for n in (o.D_A, o.D_B) for p in n.parameters()])
        if step == 2:
            o.set_epoch(150, 100, 100); m.set_epoch(150, 100, 100)            # LR x (1 - 50/101)
        if step == 3:                                                         # resume from the checkpoint written after step 2
            torch.manual_seed(999)
            mr = u.CycleGAN(n_blocks=6, dtype=torch.float32, use_graph=False, pool_size=3, pool_seed=1234)
            mr.load(str(tmp_path / "ckpt.pt"))
        lo = o.train_step(rA, rB)
        lm = m.train_step(rA.cuda(), rB.cuda())
        check(step, lo, lm, "graph")
        if mr is not None:
            check(step, lo, mr.train_step(rA.cuda(), rB.cuda()), "resumed")
        if step == 2:
            m.save(str(tmp_path / "ckpt.pt"))
        w_or2 = torch.cat([p.detach().reshape(-1) for n in (o.D_A, o.D_B) for p in n.parameters()])
        upd[step] = (float((m.grp_D.flat - w_hip).abs().sum()), float((w_or2 - w_or).abs().sum()))
    # the decayed steps move the discriminators' weights by the same total amount as the oracle's LambdaLR'd Adam does
    # (scale 1 - 50/101: ignoring it would double the update), and by visibly less than the undecayed step before
    for step in (1, 2, 3):
        assert abs(upd[step][0] / upd[step][1] - 1.0) < 0.05, (step, upd[step])
    assert upd[2][0] < 0.75 * upd[1][0]
    assert m.pool_B.n == 3 and len(o.pool_B.images) == 3
    # the discriminators really saw pooled (older) fakes: the pools' contents agree with the oracle's slot for slot (the same
    # image of the same step in the same slot: fakes of different steps / slots differ by O(0.5); the same fake differs by the
    # accumulated weight drift of the steps before it, a few 1e-2 at most)
    for mine, theirs in ((m.pool_B, o.pool_B), (m.pool_A, o.pool_A)):
        for j in range(3):
            dlt = (u.ops.from_nhwc(mine.buf[j:j + 1], 3).cpu() - theirs.images[j]).abs()
            assert float(dlt.max()) < 0.25 and float(dlt.mean()) < 0.03, (j, float(dlt.max()), float(dlt.mean()))
    m.close(); mr.close()


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_staged_backward_step_equals_single_stage(graph):
    """The data-parallel form of the step (backward cut into 4 + 2 stages at ResBlock / layer boundaries, one HIP graph per
    stage in graph mode, gradient buckets = contiguous slices of the interleaved flat buffer) without any collective: bitwise
    the single-stage step over 3 steps - same kernels, same order, same sums."""
    import unpaired_image_generation_amd as u
    torch.manual_seed(21)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    torch.manual_seed(5)
    m0 = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=graph)
    m1 = u.CycleGAN(n_blocks=6, dtype=torch.bfloat16, use_graph=graph, stage_backward=True)
    m1.load_state_dicts(*[n.state_dict() for n in m0.nets()])
    assert len(m1.buckets_G) == 4 and len(m1.buckets_D) == 2 and len(m0.buckets_G) == 1
    assert m1.buckets_G[0][1] == m1.grp_G.flat.numel() and m1.buckets_G[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(m1.buckets_G, m1.buckets_G[1:]))
    for step in range(3):
        l0, l1 = m0.train_step(rA, rB), m1.train_step(rA, rB)
        assert l0 == l1, (step, l0, l1)
    assert torch.equal(m0.grp_G.flat, m1.grp_G.flat) and torch.equal(m0.grp_D.flat, m1.grp_D.flat)
    if graph:
        assert m1.graph_active and len(m1._graphs.g1) == 4 and len(m1._graphs.g2) == 2
    m0.close(); m1.close()


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_combined_pass_weight_gradient_equals_separate(graph, monkeypatch):
    """The generator phase's weight gradients with ONE launch per ResBlock conv pair for both generator passes
    (ops.combined_pass_wgrad, the default) against one launch per pass: the same sums in another order - fp32 partials, so the
    flat gradient buffer agrees to ~1e-6 of its scale after the first step, and the losses of the next steps stay together."""
    import unpaired_image_generation_amd as u
    ops = u.ops
    torch.manual_seed(22)
    rA, rB = (torch.rand(2, 3, 32, 256, device="cuda") * 2 - 1 for _ in range(2))     # 64-pixel-wide ResBlock maps: the image-row kernel's shape
    torch.manual_seed(6)
    monkeypatch.setattr(ops, "COMBINE_PASS_WGRAD", False)
    m0 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph, stage_backward=True)
    l0 = [m0.train_step(rA, rB) for _ in range(1)]
    g0 = m0.grp_G.grad.clone()
    l0 += [m0.train_step(rA, rB) for _ in range(2)]
    monkeypatch.setattr(ops, "COMBINE_PASS_WGRAD", True)
    torch.manual_seed(6)
    m1 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph, stage_backward=True)
    seen = []
    real = ops._combined_wgrad
    monkeypatch.setattr(ops, "_combined_wgrad", lambda *a: (seen.append(real(*a)) or seen[-1]))
    l1 = [m1.train_step(rA, rB) for _ in range(1)]
    g1 = m1.grp_G.grad.clone()
    l1 += [m1.train_step(rA, rB) for _ in range(2)]
    assert not ops._WG_STASH, "the region must be closed (and its stash flushed) after the step"
    # every generator conv pair (6 ResBlock convs, 4 stride-2 layers, stem, head) of every Python-level backward (3 eager steps;
    # warm-up + capture in graph mode): stashed on the first visit, combined on the second
    assert sum(seen) >= 2 * 12 and sum(seen) % (2 * 12) == 0, sum(seen)
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-5 * scale
    # first step: the same forward, so the losses agree to rounding; later steps drift apart the way the oracle itself does under a
    # 1e-6 perturbation of its weights (scripts/oracle_sensitivity.py: percents within a few steps), so they only have to stay close
    for i, (a, b) in enumerate(zip(l0, l1)):
        for k in a:
            assert abs(a[k] - b[k]) <= (1e-5 if i == 0 else 4e-2) * max(1.0, abs(a[k])), (i, k, a[k], b[k])
    m0.close(); m1.close()


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_side_streams_equal_single_stream(graph, monkeypatch):
    """The stream-parallel form of the step (parameter-gradient kernels on a side stream joined once per phase, generator update
    on its own stream under the discriminator phase - the defaults of round 1, now opt-in / data-parallel only because they
    measure slower on one GPU) against the single-stream default: the same kernels on the same data in another interleaving,
    so losses and parameters must agree bitwise over 3 steps."""
    import unpaired_image_generation_amd as u
    ops = u.ops
    torch.manual_seed(23)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    torch.manual_seed(7)
    monkeypatch.setattr(ops, "PARALLEL_BACKWARD", False)
    monkeypatch.setenv("UIG_OVERLAP_UPDATE", "0")
    m0 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph)
    assert not m0.overlap_update
    l0 = [m0.train_step(rA, rB) for _ in range(3)]
    monkeypatch.setattr(ops, "PARALLEL_BACKWARD", True)
    monkeypatch.setenv("UIG_OVERLAP_UPDATE", "1")
    torch.manual_seed(7)
    m2 = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph)      # same seed as m0: same initial weights
    assert m2.overlap_update
    l2 = [m2.train_step(rA, rB) for _ in range(3)]
    assert l0 == l2, (l0, l2)
    assert torch.equal(m0.grp_G.flat, m2.grp_G.flat) and torch.equal(m0.grp_D.flat, m2.grp_D.flat)
    m0.close(); m2.close()


@pytest.mark.parametrize("staged", [False, True], ids=["one-bucket-per-phase", "staged-backward"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_two_rank_data_parallel_step_equals_full_batch(graph, staged, tmp_path):
    """Two ranks (gloo, both on GPU 0; RCCL refuses two ranks per device) each train on one image of a 2-image batch through the
    product's data-parallel step - the round-3 default (one all-reduce per optimiser group behind its phase: the generators' runs
    under the discriminator phase) and the staged form (backward in stages, a gradient bucket all-reduced behind each stage, between
    the stage graphs in graph mode); both generator passes' weight gradients in one launch, 1/world folded into Adam - against ONE
    process training on the full batch.  InstanceNorm is per sample and every loss is a batch mean, so: sum of the ranks' gradients = 2 x the
    full-batch gradient, and the parameters agree after two steps; both up to the fp32 summation order.  A bucket reduced before
    its last gradient had landed would show as a missing contribution here (at world size 1 it cannot)."""
    import socket
    import subprocess
    import sys
    import unpaired_image_generation_amd as u
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_gloo2_worker.py")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    outs = [str(tmp_path / f"r{r}.pt") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(port), outs[r], "1" if graph else "0", "1" if staged else "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    # the full-batch reference, in this process, while the ranks run
    torch.manual_seed(31)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    torch.manual_seed(9)
    m = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=graph)
    m.train_step(rA, rB)
    gG, gD = m.grp_G.grad.clone().cpu(), m.grp_D.grad.clone().cpu()
    m.train_step(rA, rB)
    pG, pD = m.grp_G.flat.cpu(), m.grp_D.flat.cpu()
    m.close()
    for p in procs:
        out, _ = p.communicate(timeout=600)
        print(out[-1500:])
        assert p.returncode == 0 and "DP_GLOO2_OK" in out
    r0, r1 = (torch.load(o, weights_only=True) for o in outs)
    for k in ("gG", "gD", "pG", "pD"):
        assert torch.equal(r0[k], r1[k]), f"ranks disagree on {k}"          # both hold the reduced buffers / identical replicas
    for k, ref in (("gG", gG), ("gD", gD)):
        d = (r0[k] * 0.5 - ref).abs().max()
        assert float(d) <= 2e-4 * float(ref.abs().max()), (k, float(d), float(ref.abs().max()))
    for k, ref in (("pG", pG), ("pD", pD)):
        d = (r0[k] - ref).abs()
        # Adam normalises the step: a gradient element within rounding of zero can flip its 2e-4 step, so bound the mean tightly and the max by one step
        assert float(d.mean()) <= 2e-6 and float(d.max()) <= 5e-4, (k, float(d.mean()), float(d.max()))


def test_process_group_lifecycle_then_new_graph_model():
    """Round 2 saw `Fatal Python error: Segmentation fault` inside CUDAGraph.replay: the first graph replay of a freshly captured
    model after an in-process init_process_group('nccl') ... CycleGAN.close() ... destroy_process_group().  The whole sequence -
    group up, graph steps with the forced RCCL exchange, close(), group down, NEW graph model trained, twice over - in ONE
    process (tests/_pg_lifecycle_worker.py, faulthandler on; a worker so that a native fault fails this test with its traceback
    instead of taking the pytest session down)."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pg_lifecycle_worker.py")
    r = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:]); print(r.stderr[-6000:])
    assert r.returncode == 0, f"worker exit status {r.returncode}"
    assert "PG_LIFECYCLE_OK" in r.stdout


def test_generator_hypothesis_odd_shapes_fp32():
    """SURVEY §4's odd-shape tier: hypothesis draws batch 1-3 and arbitrary H, W in [12, 88] (odd sizes, non-multiples of the
    tile and of 4, non-square) - generator forward L-inf < 1e-3 and every weight gradient (relative L2 < 1e-2) against the
    oracle on the exact-f32 path.  Sizes that are not multiples of 4 come back as 4*ceil(ceil(H/2)/2), as the stock stride-2 /
    output_padding-1 chain returns them (asserted through the oracle's own output shape).  Derandomised: the same examples
    every run."""
    import unpaired_image_generation_amd as u
    from hypothesis import given, settings, strategies as st, HealthCheck
    from oracle.torch_oracle import Generator as OG, init_weights
    torch.manual_seed(77)
    og = init_weights(OG(n_blocks=2))
    g = u.Generator(n_blocks=2, dtype=torch.float32)
    g.load_state_dict(og.state_dict())
    seen = []

    @settings(max_examples=14, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(B=st.integers(1, 3), H=st.integers(12, 88), W=st.integers(12, 88), seed=st.integers(0, 10 ** 6))
    def run(B, H, W, seed):
        gen = torch.Generator().manual_seed(seed)
        x = torch.rand(B, 3, H, W, generator=gen) * 2 - 1
        og.zero_grad(); g.zero_grad()
        yr = og(x)
        t = torch.randn(yr.shape, generator=gen)
        (yr * t).sum().backward()
        y = g(x.cuda())
        assert tuple(y.shape) == tuple(yr.shape), (tuple(y.shape), tuple(yr.shape))
        linf = float((y.detach().cpu() - yr.detach()).abs().max())
        assert linf < 1e-3, (B, H, W, linf)
        (y * t.cuda()).sum().backward()
        ref = dict(og.named_parameters())
        for k, p in g.named_parameters():
            if k.endswith(".weight"):
                r = ref[k].grad
                rel = float((p.grad.cpu() - r).norm() / (r.norm() + 1e-30))
                assert rel < 1e-2, (B, H, W, k, rel)
        seen.append((B, H, W))

    run()
    print("hypothesis shapes:", seen)
    assert len(seen) >= 10 and any(h % 4 or w % 4 for _, h, w in seen)


def test_graph_step_with_rccl_exchange_world1_and_close():
    """Guards two aborts seen in round 1 (graph capture with a live process group; process exit with graphs + RCCL alive) and is the
    standing probe for round 2's fault (`Segmentation fault` inside CUDAGraph.replay: the first replay of a fresh model after an
    in-process init_process_group('nccl') ... CycleGAN.close() ... destroy_process_group()).
    init_process_group('nccl', world_size=1), CycleGAN(use_graph=True, force_exchange=True) - default and staged form - so that the
    RCCL all-reduces really run between the graph replays; 3 steps bitwise equal to the model without exchange (a 1-rank sum is the
    identity); then the ordered teardown - CycleGAN.close(), destroy_process_group().
    Round 4: IN THIS PROCESS by default (the round-2 arrangement: the group is created and destroyed inside the pytest session and
    every later test file captures and replays graphs behind it), faulthandler on, placed last in this file.  Round 3 removed what is
    believed to have caused the fault (tensors of the graphs' private memory pool outliving their graphs: DESIGN.md §4) and three
    runs of this arrangement were clean; if it ever faults again the native frames of THAT run are the evidence to work from.
    UIG_TEST_PG_WORKER=1: the round-2/3 fallback (one worker process = one rank whose group lives until the process exits; its exit
    status is checked: tests/_rccl_world1_worker.py)."""
    import faulthandler
    import subprocess
    import sys
    if os.environ.get("UIG_TEST_PG_WORKER") == "1":
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_world1_worker.py")
        r = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
        print(r.stdout[-2000:]); print(r.stderr[-3000:])
        assert r.returncode == 0, f"worker exit status {r.returncode}"
        assert "RCCL_WORLD1_OK" in r.stdout
        return
    import importlib
    import torch.distributed as dist
    faulthandler.enable(all_threads=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    importlib.import_module("_rccl_world1_worker").main()
    assert not dist.is_initialized()
    import unpaired_image_generation_amd as u
    # a NEW graph model right behind the teardown, in this process: capture + three replays
    torch.manual_seed(2)
    rA, rB = (torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1 for _ in range(2))
    m = u.CycleGAN(n_blocks=3, dtype=torch.bfloat16, use_graph=True)
    ls = [m.train_step(rA, rB) for _ in range(3)]
    assert m.graph_active and all(v == v for l in ls for v in l.values())
    m.close()
