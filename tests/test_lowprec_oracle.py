"""CPU checks of the same-rounding low-precision emulation (oracle/lowprec_oracle.py) that the -m gpu tests use as the
checker for the bf16 (configs[1]) and MX-fp8 (configs[4]) steps: it must be the fp32 oracle plus rounding - same weights for
the same seed, same step order - and its storage points must behave as stated."""
import torch
import torch.nn.functional as F

from oracle import lowprec_oracle as LP
from oracle import mx_fp8 as M
from oracle.torch_oracle import CycleGANOracle


def test_store_rounds_value_and_gradient_operand_copy_passes_gradient():
    x = (torch.randn(64) * 3).requires_grad_(True)
    y = LP.store(x)
    assert torch.equal(y, x.detach().to(torch.bfloat16).float()) and torch.equal(LP.store(y), y)
    g = torch.randn(64)
    y.backward(g)
    assert torch.equal(x.grad, g.to(torch.bfloat16).float())
    w = torch.randn(64, requires_grad=True)
    LP.wop(w).backward(g)
    assert torch.equal(w.grad, g)


def test_mx_conv_function_matches_its_parts():
    """forward = oracle/mx_fp8.conv3x3_mx_forward; input gradient = fp8 main term + exact bf16 border terms; weight / bias
    gradients = the bf16-operand convolution's (what tests/test_fp8_gpu.py checks the device against, here as one Function)"""
    torch.manual_seed(3)
    bf = LP._bf
    x = bf(torch.randn(2, 64, 9, 10)).requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3) * 0.05).requires_grad_(True)
    b = torch.randn(64, requires_grad=True)
    dy = bf(torch.randn(2, 64, 9, 10))
    y = LP._MXConv3x3.apply(x, w, b)
    assert torch.equal(y, M.conv3x3_mx_forward(x.detach(), w.detach(), b.detach(), True))
    y.backward(dy)
    xr = x.detach().clone().requires_grad_(True)
    wr = bf(w.detach()).requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br).backward(dy)
    assert torch.allclose(w.grad, wr.grad, rtol=1e-5, atol=1e-5) and torch.allclose(b.grad, br.grad, rtol=1e-5, atol=1e-5)
    zp = F.conv_transpose2d(dy, wr.detach(), None, 1, 1)
    want = M.conv3x3_mx_dgrad_zero_pad(dy, w.detach()) + (xr.grad - zp)
    assert torch.allclose(x.grad, want, rtol=1e-5, atol=1e-5)
    # and the quantised gradient stays close to the exact one (3-bit mantissas, 576-term sums)
    assert float((x.grad - xr.grad).abs().max()) < 8e-2 * float(xr.grad.abs().max())


def test_lowprec_steps_track_the_fp32_oracle():
    """same seed -> same weights; the bf16 emulation's first-step losses within 5 % of the fp32 oracle's, the fp8 one's within
    10 % (stated low-precision drifts; the adversarial terms are means over only 36 patch logits at 64x64), and the
    same computation otherwise: a ResBlock and a PatchGAN weight gradient within 40 % / 80 % relative L2 (cosine > 0.7) of the fp32 ones.  (That
    loose: a bf16 rounding moves a pre-activation by up to 0.4 %, which flips the ReLU mask of the ~0.3 % of elements that close
    to zero in every layer, and each flip re-routes that element's whole gradient: measured here 24 % on the ResBlock conv.  The
    DEVICE differs from this emulation only by fp32 summation order, i.e. by the rare element that lands on the other side of a
    bf16 rounding boundary: percent-level, which is what tests/test_model_gpu.py / test_fp8_gpu.py state.)"""
    torch.set_num_threads(8)
    rA, rB = torch.rand(1, 3, 64, 64) * 2 - 1, torch.rand(1, 3, 64, 64) * 2 - 1
    res = {}
    for name, mk in (("f32", lambda: CycleGANOracle(n_blocks=2)), ("bf16", lambda: LP.LowPrecOracle(n_blocks=2)),
                     ("fp8", lambda: LP.LowPrecOracle(n_blocks=2, fp8=True))):
        torch.manual_seed(5)
        o = mk()
        res[name] = (o.train_step(rA, rB), o.G_A[10].b[5].weight.grad.clone(), o.D_B[8].weight.grad.clone(), o.last["fake_B"])
    for k, v in res["f32"][0].items():
        assert abs(res["bf16"][0][k] - v) <= 5e-2 * max(1.0, abs(v)), (k, v, res["bf16"][0][k])
        assert abs(res["fp8"][0][k] - v) <= 1e-1 * max(1.0, abs(v)), (k, v, res["fp8"][0][k])
    for name, tol in (("bf16", 0.40), ("fp8", 0.80)):      # fp8 on one 16x16-map image: 61 % (cosine 0.8) measured on the ResBlock conv
        for j in (1, 2):
            ref = res["f32"][j]
            assert float((res[name][j] - ref).norm() / ref.norm()) < tol, (name, j)
            assert float(F.cosine_similarity(res[name][j].flatten(), ref.flatten(), dim=0)) > 0.7, (name, j)
        assert float((res[name][3] - res["f32"][3]).abs().max()) < (0.12 if name == "bf16" else 0.4)      # tanh output, 2-block generator on 16x16 maps
    # the stored outputs are bf16-representable
    assert torch.equal(res["bf16"][3], res["bf16"][3].to(torch.bfloat16).float())


def test_mirror_pixel_dgrad_algebra_equals_reflect_pad_gradient():
    """oracle/mx_fp8.conv3x3_mx_dgrad_reflect_mirror (the emulation of the one-launch fp8 input gradient) with quantisation switched
    off must BE the gradient of ReflectionPad2d(1)+Conv2d: pins the mirror-pixel placement (which taps read which sums) on the CPU."""
    import torch.nn.functional as F
    from oracle import mx_fp8 as M
    torch.manual_seed(11)
    for H, W in ((8, 12), (5, 4), (16, 64)):
        x = torch.randn(2, 32, H, W, requires_grad=True)
        w = torch.randn(64, 32, 3, 3)
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        dy = torch.randn_like(y)
        (g,) = torch.autograd.grad(y, x, dy)
        d = M.conv3x3_mx_dgrad_reflect_mirror(dy, w, quant=False)
        assert (d - g).abs().max() <= 1e-5 * g.abs().max()
        # quantised: close to it (fp8 noise), and not the zero-pad gradient
        dq = M.conv3x3_mx_dgrad_reflect_mirror(dy, w)
        assert (dq - g).norm() <= 0.08 * g.norm()
