"""Data-parallel path on CPU (gloo, world_size 2; runs without a GPU).

Property: InstanceNorm is per sample, so DP is exactly batch-separable -
  mean over ranks of the per-rank gradients  ==  the single-process gradient on the concatenated batch,
and after FlatGroup + GradExchange + Adam(grad_scale = 1/world) every rank holds the same parameters as a single
process trained on the whole batch.  The nets here are the CPU oracle's (stock torch); what is under test is the
product's DP plumbing: flat buffers, the asynchronous exchange ordering and the 1/world scaling."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(seed):
    from oracle.torch_oracle import Discriminator, Generator, init_weights
    torch.manual_seed(seed)
    return init_weights(Generator(n_blocks=1)), init_weights(Discriminator())


def _g_loss(G, D, x):
    import torch.nn.functional as F
    fake = G(x)
    p = D(fake)
    return F.l1_loss(fake, x) * 10 + F.mse_loss(p, torch.ones_like(p))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from unpaired_image_generation_amd.dp import FlatGroup, GradExchange
        G, D = _build(3)
        grp_G, grp_D = FlatGroup((G,), "cpu"), FlatGroup((D,), "cpu")
        xchg = GradExchange()
        assert xchg.world == world
        xchg.broadcast(grp_G.flat, 0); xchg.broadcast(grp_D.flat, 0)
        torch.manual_seed(100)
        x_all = torch.rand(world, 3, 32, 32) * 2 - 1
        x = x_all[rank:rank + 1]
        # generator phase on this rank's shard, then the same ordering as the product step:
        # start G exchange -> run the discriminator phase while it is in flight -> wait -> start D exchange -> Adam
        grp_G.zero_grad(); grp_D.set_requires_grad(False)
        _g_loss(G, D, x).backward()
        grp_D.set_requires_grad(True)
        h_g = xchg.start(grp_G.grad)
        grp_D.zero_grad()
        p = D(G(x).detach())
        ((p - 0.0) ** 2).mean().backward()
        xchg.wait(h_g)
        h_d = xchg.start(grp_D.grad)
        gG = grp_G.grad / world
        xchg.wait(h_d)
        gD = grp_D.grad / world
        if rank == 0:
            torch.save({"gG": gG.clone(), "gD": gD.clone(), "x": x_all}, out)
        # both ranks must end with identical buffers
        chk = torch.stack([gG.double().sum(), gD.double().sum()])
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(torch.equal(lst[0], t) for t in lst)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_gradient_mean_equals_full_batch(tmp_path):
    world, out = 2, str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    from unpaired_image_generation_amd.dp import FlatGroup
    G, D = _build(3)
    grp_G, grp_D = FlatGroup((G,), "cpu"), FlatGroup((D,), "cpu")
    x = got["x"]
    grp_D.set_requires_grad(False)
    _g_loss(G, D, x).backward()
    grp_D.set_requires_grad(True)
    grp_D.zero_grad()
    p = D(G(x).detach())
    ((p - 0.0) ** 2).mean().backward()
    for name, a, b in (("G", got["gG"], grp_G.grad), ("D", got["gD"], grp_D.grad)):
        rel = float((a - b).norm() / b.norm())
        assert rel < 1e-5, (name, rel)


def test_flat_group_views_and_world1_noop():
    from unpaired_image_generation_amd.dp import FlatGroup, GradExchange
    G, D = _build(5)
    ref = {k: v.clone() for k, v in G.state_dict().items()}
    grp = FlatGroup((G,), "cpu")
    assert all(torch.equal(v, ref[k]) for k, v in G.state_dict().items())      # flattening keeps the values
    assert grp.flat.numel() >= sum(p.numel() for p in G.parameters())
    for p in G.parameters():                                                   # 16-byte aligned views into the flat buffers
        assert p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0
        assert grp.flat.data_ptr() <= p.data_ptr() < grp.flat.data_ptr() + grp.flat.numel() * 4
    G(torch.rand(1, 3, 32, 32)).sum().backward()
    assert float(grp.grad.abs().sum()) > 0                                      # autograd accumulated into the flat buffer
    grp.flat.add_(1.0)
    assert torch.equal(next(G.parameters()).data, ref[next(iter(ref))] + 1.0)  # updating the flat buffer updates the module
    x = GradExchange()
    assert x.world == 1 and x.start(grp.grad) is None
    x.wait(None)
