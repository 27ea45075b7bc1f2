"""Host-side utilities next to the hot path: image history pool and LR schedule.  CPU only."""
import torch

import unpaired_image_generation_amd as u


def test_linear_decay_scale_matches_the_recipe():
    f = u.linear_decay_scale
    assert f(0) == 1.0 and f(100) == 1.0
    assert abs(f(101) - (1 - 1 / 101)) < 1e-12
    assert abs(f(200) - (1 - 100 / 101)) < 1e-12 and f(200) > 0
    assert abs(f(150, 100, 100) - (1 - 50 / 101)) < 1e-12
    assert f(5, 0, 9) == 0.5


def test_image_pool_disabled_is_identity():
    p = u.ImagePool(0)
    x = torch.arange(12.0).view(3, 4)
    assert p.query(x) is x
    out = torch.zeros_like(x)
    assert p.query(x, out) is out and torch.equal(out, x)


def test_image_pool_fill_then_swap_semantics():
    p = u.ImagePool(4, seed=3)
    imgs = [torch.full((2, 3), float(i)) for i in range(40)]
    # while filling, every image comes straight back and is stored
    for i in range(2):
        out = p.query(torch.stack(imgs[2 * i:2 * i + 2]))
        assert torch.equal(out, torch.stack(imgs[2 * i:2 * i + 2]))
    assert p.n == 4 and sorted(float(v[0, 0]) for v in p.buf) == [0.0, 1.0, 2.0, 3.0]
    seen_old = seen_new = 0
    stored = {0.0, 1.0, 2.0, 3.0}
    for i in range(4, 40):
        out = p.query(imgs[i].unsqueeze(0))
        v = float(out[0, 0, 0])
        assert torch.equal(out[0], torch.full((2, 3), v))       # whole images move, never mixtures
        if v == float(i):
            seen_new += 1                                        # returned as is: the pool is unchanged
        else:
            seen_old += 1
            assert v in stored                                   # an OLD image came back ...
            stored.remove(v); stored.add(float(i))               # ... and the new one took its slot
        assert {float(b[0, 0]) for b in p.buf} == stored
    assert seen_old > 5 and seen_new > 5                         # both branches of the coin are exercised


def test_image_pool_is_reproducible_and_checkpointable():
    a, b = u.ImagePool(3, seed=11), u.ImagePool(3, seed=11)
    xs = [torch.randn(2, 5) for _ in range(12)]
    for x in xs[:6]:
        assert torch.equal(a.query(x), b.query(x))
    c = u.ImagePool(3, seed=99)
    c.load_state_dict(a.state_dict())
    for x in xs[6:]:
        assert torch.equal(a.query(x), c.query(x))


def test_oracle_pool_and_schedule_restate_the_product():
    """The oracle's pool (stdlib uniform / randint, as the public recipe writes it) consumes its RNG exactly like the product's
    pool (random / randrange), so with equal seeds both make the same decisions; its LambdaLR schedule gives the product's
    multiplier.  This is what lets tests/test_model_gpu.py compare a pooled, scheduled run step by step."""
    from oracle.torch_oracle import CycleGANOracle, OraclePool
    a, b = u.ImagePool(3, seed=5), OraclePool(3, 5)
    for i in range(30):
        x = torch.full((2, 4), float(i)) + torch.tensor([[0.0], [0.5]])
        assert torch.equal(a.query(x), b.query(x)), i
    o = CycleGANOracle(n_blocks=1)
    for epoch in (1, 100, 101, 150, 200):
        o.set_epoch(epoch, 100, 100)
        want = 2e-4 * u.linear_decay_scale(epoch, 100, 100)
        assert all(abs(g["lr"] - want) < 1e-18 for opt in (o.opt_G, o.opt_D) for g in opt.param_groups), epoch
