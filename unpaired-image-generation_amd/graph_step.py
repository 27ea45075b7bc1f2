"""HIP-graph execution of the train step: the static-shape, sync-free step is captured once into four graphs
(G fwd+bwd | D fwd+bwd | Adam G + repack G | Adam D + repack D) and replayed; data-parallel all-reduces are enqueued
between the replays on the communication stream and the generator update graph is replayed on its own stream, so the
generator exchange, Adam and weight repack all run under the discriminators' forward+backward.
Replaces a tracing compiler: one capture of the hand-written kernel sequence, no per-op host overhead afterwards."""
from __future__ import annotations

import torch

from . import ops


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

    st.g1, st.g2, st.g3, st.g4 = (torch.cuda.CUDAGraph() for _ in range(4))
    # capture_error_mode thread_local: with a process group alive, RCCL's watchdog thread polls its events (hipEventQuery)
    # while this thread captures; under the default global mode that poll is an illegal call DURING CAPTURE and aborts the
    # process (seen in 5 of 8 launches under torchrun).
    cem = dict(capture_error_mode="thread_local")
    with torch.cuda.graph(st.g1, **cem):
        st.xa, st.xb = model.to_phys(st.real_A), model.to_phys(st.real_B)
        st.fake_B, st.fake_A, lg = model._g_phase(st.xa, st.xb)
        st.lg = torch.cat([l.detach() for l in lg])
    pool = st.g1.pool()
    # the discriminators read their fakes from static buffers: the image pools (if enabled) fill them between the replays
    st.pooled = model.pool_B.size > 0
    st.dfake_B = torch.empty_like(st.fake_B) if st.pooled else st.fake_B
    st.dfake_A = torch.empty_like(st.fake_A) if st.pooled else st.fake_A
    with torch.cuda.graph(st.g2, pool=pool, **cem):
        ld = model._d_phase(st.xa, st.xb, st.dfake_B, st.dfake_A)
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
    st.g1.replay()
    h_g = model.xchg.start(model.grp_G.grad)      # exchange, Adam and repack of the generators run under the discriminator graph
    if model.overlap_update:
        upd.wait_stream(main)
        with torch.cuda.stream(upd):
            model.xchg.wait(h_g, model.device)
            st.g3.replay()
    if st.pooled:
        model._pool_fakes(st.fake_B, st.fake_A, st.dfake_B, st.dfake_A)
    st.g2.replay()
    if not model.overlap_update:
        model.xchg.wait(h_g, model.device)
        st.g3.replay()
    h_d = model.xchg.start(model.grp_D.grad)
    model.xchg.wait(h_d, model.device)
    st.g4.replay()
    main.wait_stream(upd)
    model.grp_G.step += 1
    model.grp_D.step += 1
    return st.losses
