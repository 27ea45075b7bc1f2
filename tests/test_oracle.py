"""Pin the CPU oracle: SURVEY.md Appendix B known answers + the independent fp64 restatement."""
import hashlib
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import naive_fp64 as nv
from oracle.torch_oracle import CycleGANOracle, Discriminator, Generator, init_weights


def test_param_counts_and_state_dict_layout():
    g9, g6, d = Generator(n_blocks=9), Generator(n_blocks=6), Discriminator()
    assert sum(p.numel() for p in g9.parameters()) == 11_378_179
    assert sum(p.numel() for p in g6.parameters()) == 7_837_699
    assert sum(p.numel() for p in d.parameters()) == 2_764_737
    assert (len(g6.state_dict()), len(g9.state_dict()), len(d.state_dict())) == (36, 48, 10)
    assert "10.b.1.weight" in g9.state_dict() and "1.weight" in g9.state_dict()


def test_config1_known_answer():
    """Appendix B recipe B1 (BASELINE.json configs[0]): G(6) forward on 1x3x64x64, seed 1234."""
    torch.manual_seed(1234)
    g = init_weights(Generator(n_blocks=6))
    x = torch.rand(1, 3, 64, 64) * 2 - 1
    with torch.no_grad():
        y = g(x)
    assert hashlib.sha256(x.numpy().tobytes()).hexdigest()[:16] == "66515c42021d0329"
    ref = torch.tensor([0.6002455353736877, -0.56021648645401, -0.49123692512512207])
    assert torch.allclose(y[0, :, 0, 0], ref, atol=1e-5)       # tolerance gate is authoritative across hosts
    assert abs(y.mean().item() - (-0.09294009953737259)) < 1e-5
    assert abs(y.std().item() - 0.4873424768447876) < 1e-5
    gold = np.load("tests/golden/config1_g6_64.npz")
    assert np.array_equal(gold["x"], x.numpy())
    assert np.abs(gold["y"] - y.numpy()).max() < 1e-5


@pytest.mark.timeout(300)
def test_train_step_known_answer():
    """Appendix B recipe B2: the 8 first-step losses, seed 0, B=1, 256x256."""
    torch.manual_seed(0)
    o = CycleGANOracle()
    rA = torch.rand(1, 3, 256, 256) * 2 - 1
    rB = torch.rand(1, 3, 256, 256) * 2 - 1
    got = o.train_step(rA, rB)
    want = dict(idt_A=3.4785473346710205, idt_B=3.214613437652588, G_A=1.4040476083755493, G_B=1.788599967956543,
                cyc_A=6.421139240264893, cyc_B=6.964799880981445, D_A=1.9957597255706787, D_B=1.6433069705963135)
    for k, v in want.items():
        assert abs(got[k] - v) < 2e-4 * max(1.0, abs(v)), (k, got[k], v)


# ---- torch CPU ops vs the independent fp64 restatement (small shapes) -------------------------
def _r(*s, seed=0):
    return np.random.default_rng(seed).standard_normal(s)


@pytest.mark.parametrize("k,s,p", [(3, 1, 0), (3, 2, 1), (4, 2, 1), (4, 1, 1), (7, 1, 0)])
def test_conv2d_fwd_bwd_vs_fp64(k, s, p):
    x, w, b = _r(2, 5, 11, 12), _r(6, 5, k, k, seed=1), _r(6, seed=2)
    xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
    y = F.conv2d(xt, wt, bt, s, p)
    assert np.abs(y.detach().numpy() - nv.conv2d(x, w, b, s, p)).max() < 1e-10
    dy = _r(*y.shape, seed=3)
    y.backward(torch.tensor(dy))
    dx, dw, db = nv.conv2d_bwd(dy, x, w, s, p)
    assert np.abs(xt.grad.numpy() - dx).max() < 1e-10
    assert np.abs(wt.grad.numpy() - dw).max() < 1e-10
    assert np.abs(bt.grad.numpy() - db).max() < 1e-10


def test_conv_transpose_vs_fp64():
    x, w, b = _r(2, 4, 5, 6), _r(4, 3, 3, 3, seed=1), _r(3, seed=2)
    y = F.conv_transpose2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), 2, 1, 1)
    assert y.shape[-2:] == (10, 12)
    assert np.abs(y.numpy() - nv.conv_transpose2d(x, w, b)).max() < 1e-10


@pytest.mark.parametrize("p", [1, 3])
def test_reflection_pad_vs_fp64(p):
    x = _r(2, 3, 7, 8)
    xt = torch.tensor(x, requires_grad=True)
    y = F.pad(xt, (p, p, p, p), mode="reflect")
    assert np.array_equal(y.detach().numpy(), nv.reflection_pad2d(x, p))
    dy = _r(*y.shape, seed=4)
    y.backward(torch.tensor(dy))
    assert np.abs(xt.grad.numpy() - nv.reflection_pad2d_bwd(dy, p)).max() < 1e-12


def test_instance_norm_fwd_bwd_vs_fp64():
    x = _r(2, 3, 9, 10) * 3 + 1
    xt = torch.tensor(x, requires_grad=True)
    y = F.instance_norm(xt, eps=1e-5)
    assert np.abs(y.detach().numpy() - nv.instance_norm(x)).max() < 1e-10
    dy = _r(*x.shape, seed=5)
    y.backward(torch.tensor(dy))
    assert np.abs(xt.grad.numpy() - nv.instance_norm_bwd(dy, x)).max() < 1e-10


def test_losses_and_adam_vs_fp64():
    a, b = _r(2, 3, 8, 8), _r(2, 3, 8, 8, seed=1)
    assert abs(F.l1_loss(torch.tensor(a), torch.tensor(b)).item() - nv.l1_loss(a, b)) < 1e-12
    assert abs(F.mse_loss(torch.tensor(a), torch.ones(a.shape, dtype=torch.float64)).item() - nv.mse_const(a, 1.0)) < 1e-12
    p = torch.tensor(_r(50), requires_grad=True)
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999))
    pn, m, v = p.detach().numpy().copy(), np.zeros(50), np.zeros(50)
    for step in (1, 2, 3):
        g = _r(50, seed=10 + step)
        p.grad = torch.tensor(g); opt.step()
        pn, m, v = nv.adam_step(pn, g, m, v, step)
        assert np.abs(p.detach().numpy() - pn).max() < 1e-12
