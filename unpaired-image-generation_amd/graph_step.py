"""HIP-graph execution of the train step: the static-shape, sync-free step is captured once into graphs
(G fwd+bwd | D fwd+bwd | Adam G + repack G | Adam D + repack D) and replayed.  Data parallel: a group's all-reduce is enqueued
on the communication stream between the replays (default since round 3: one per optimiser group behind its phase graph).  In
the staged form (stage_backward / UIG_DP_STAGED=1) each phase's backward pass is cut into stages (cyclegan._g_stages /
_d_stages) and every stage is its own graph: bucket k's all-reduce is enqueued between the replays of stage k and stage k+1 and
runs under the latter; with UIG_OVERLAP_UPDATE=1 the generator update graph (wait for the buckets, Adam, weight repack) is
replayed on its own stream under the discriminators' forward+backward.
Replaces a tracing compiler: one capture of the hand-written kernel sequence, no per-op host overhead afterwards."""
from __future__ import annotations

import torch

from . import ops
from .dp import run_exchange_phase


class _Captured:
    pass


def _capture(model, real_A, real_B):
    st = _Captured()
    dev = model.device
    st.real_A = torch.empty_like(real_A, device=dev).copy_(real_A)
    st.real_B = torch.empty_like(real_B, device=dev).copy_(real_B)
    for grp in (model.grp_G, model.grp_D):
        if not hasattr(grp, "state16"):
            grp.state16 = ops.new_adam_state(dev, grp.step, model.lr_scale)
        grp.state16[0] = grp.step

    # one eager warm-up step on a side stream (lazy kernel attributes, allocator warm-up), with all training state restored
    saved = [t.clone() for g in (model.grp_G, model.grp_D) for t in (g.flat, g.m, g.v)]
    steps = (model.grp_G.step, model.grp_D.step)
    pools = (model.pool_B.state_dict(), model.pool_A.state_dict())
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model._step_eager(model.to_phys(st.real_A), model.to_phys(st.real_B))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize(dev)
    it = iter(saved)
    for g in (model.grp_G, model.grp_D):
        for t in (g.flat, g.m, g.v):
            t.copy_(next(it))
    model.grp_G.step, model.grp_D.step = steps
    model.pool_B.load_state_dict(pools[0]); model.pool_A.load_state_dict(pools[1])
    model.repack()                                   # the warm-up step left the kernel operands at its own updated weights

    st.g3, st.g4 = (torch.cuda.CUDAGraph() for _ in range(2))
    st.g1 = [torch.cuda.CUDAGraph() for _ in model.buckets_G]      # one graph per backward stage (one in all when no collective runs)
    st.g2 = [torch.cuda.CUDAGraph() for _ in model.buckets_D]
    # capture_error_mode thread_local: with a process group alive, RCCL's watchdog thread polls its events (hipEventQuery)
    # while this thread captures; under the default global mode that poll is an illegal call DURING CAPTURE and aborts the
    # process (seen in 5 of 8 launches under torchrun).
    cem = dict(capture_error_mode="thread_local")
    rg, rd = {}, {}
    pool = None
    gen = None
    for k, g in enumerate(st.g1):
        with torch.cuda.graph(g, **(dict(pool=pool) if pool is not None else {}), **cem):
            if k == 0:
                st.xa, st.xb = model.to_phys(st.real_A), model.to_phys(st.real_B)
                gen = model._g_stages(st.xa, st.xb, rg)
            assert next(gen) == k
            if k == len(st.g1) - 1:
                assert next(gen, None) is None                 # run the generator's tail (un-freeze D) inside the capture
                st.fake_B, st.fake_A = rg["fake_B"], rg["fake_A"]
                st.lg = torch.cat([l.detach() for l in rg["losses"]])
        pool = st.g1[0].pool()
    # the discriminators read their fakes from static buffers: the image pools (if enabled) fill them between the replays
    st.pooled = model.pool_B.size > 0
    st.dfake_B = torch.empty_like(st.fake_B) if st.pooled else st.fake_B
    st.dfake_A = torch.empty_like(st.fake_A) if st.pooled else st.fake_A
    for k, g in enumerate(st.g2):
        with torch.cuda.graph(g, pool=pool, **cem):
            if k == 0:
                gen = model._d_stages(st.xa, st.xb, st.dfake_B, st.dfake_A, rd)
            assert next(gen) == k
            if k == len(st.g2) - 1:
                assert next(gen, None) is None
                ld = rd["losses"]
                st.losses = torch.cat([st.lg, ld[0][0].detach() + ld[0][1].detach(), ld[1][0].detach() + ld[1][1].detach()])
    with torch.cuda.graph(st.g3, pool=pool, **cem):
        g = model.grp_G
        ops.adam_flat_graph(g.flat, g.grad, g.m, g.v, model.lr, model.b1, model.b2, model.eps, g.state16, 1.0 / model.world)
        model._packer_of("G").run()
    with torch.cuda.graph(st.g4, pool=pool, **cem):
        g = model.grp_D
        ops.adam_flat_graph(g.flat, g.grad, g.m, g.v, model.lr, model.b1, model.b2, model.eps, g.state16, 1.0 / model.world)
        model._packer_of("D").run()
    return st


def graph_train_step(model, real_A, real_B):
    st = model._graphs
    if st is None or st.real_A.shape != real_A.shape or st.real_B.shape != real_B.shape or st.real_A.dtype != real_A.dtype:
        try:
            st = model._graphs = _capture(model, real_A, real_B)
        except RuntimeError as e:
            # Capture is an optimisation, not a requirement: if the runtime refuses it (e.g. another thread touched the
            # device mid-capture), say so loudly once and run the identical kernel sequence eagerly from now on.
            import sys
            print(f"[uig] HIP-graph capture failed ({e}); continuing in eager mode", file=sys.stderr, flush=True)
            torch.cuda.synchronize(model.device)
            model._graphs, model.use_graph = None, False
            for grp in (model.grp_G, model.grp_D):          # the eager Adam counts steps on the host: drop the device records
                if hasattr(grp, "state16"):
                    del grp.state16
            return model._step_eager(model.to_phys(real_A), model.to_phys(real_B))
    st.real_A.copy_(real_A, non_blocking=True)
    st.real_B.copy_(real_B, non_blocking=True)
    main, upd = torch.cuda.current_stream(model.device), model._update_stream()
    def replay_stages(graphs):
        for k, g in enumerate(graphs):
            g.replay()
            yield k

    # bucket k's all-reduce is enqueued right behind stage k's graph and runs under stage k+1's
    h_g = run_exchange_phase(replay_stages(st.g1), model.xchg, model.grp_G.grad, model.buckets_G)
    if model.overlap_update:      # exchange wait, Adam and repack of the generators run under the discriminator graphs
        upd.wait_stream(main)
        with torch.cuda.stream(upd):
            model.xchg.wait_all(h_g, model.device)
            st.g3.replay()
    if st.pooled:
        model._pool_fakes(st.fake_B, st.fake_A, st.dfake_B, st.dfake_A)
    h_d = run_exchange_phase(replay_stages(st.g2), model.xchg, model.grp_D.grad, model.buckets_D)
    if not model.overlap_update:
        model.xchg.wait_all(h_g, model.device)
        st.g3.replay()
    model.xchg.wait_all(h_d, model.device)
    st.g4.replay()
    main.wait_stream(upd)
    model.grp_G.step += 1
    model.grp_D.step += 1
    return st.losses
