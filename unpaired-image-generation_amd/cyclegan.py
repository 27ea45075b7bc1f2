"""CycleGAN train step (layer L4) on the HIP operator path, single GPU or data parallel.

`CycleGAN.train_step(real_A, real_B)` performs exactly the canonical optimisation step:
  fake_B=G_A(A) rec_A=G_B(fake_B) fake_A=G_B(B) rec_B=G_A(fake_A) idt_A=G_A(B) idt_B=G_B(A);
  loss_G = MSE(D_A(fake_B),1)+MSE(D_B(fake_A),1) + lam*L1(rec_A,A)+lam*L1(rec_B,B) + lam*idt*(L1(idt_A,B)+L1(idt_B,A));
  Adam(G);  loss_D_X = 0.5*(MSE(D_X(real),1)+MSE(D_X(fake.detach()),0));  Adam(D).
MI355X-first choices (all result-preserving because InstanceNorm statistics are per sample):
  * the two passes that share a generator's weights and have no data dependency (fake and identity) run as ONE batch-2B
    launch sequence (fills the 256 CUs at small per-GPU batch); the discriminators see [real; fake] as one batch;
  * parameters, gradients and Adam moments of each optimiser group live in ONE flat fp32 buffer: Adam is a single
    kernel and the data-parallel exchange a single RCCL all-reduce per group;
  * the whole step is static-shape and sync-free, so it is captured once into HIP graphs and replayed
    (segments: G fwd+bwd | D fwd+bwd | Adam G | Adam D), with the gradient all-reduces enqueued between segments on a
    communication stream.  Data parallel, default (round 3): ONE all-reduce per optimiser group right behind its phase - the
    generators' runs under the whole discriminator phase, the discriminators' under the generators' Adam + weight repack, all
    on the main stream.  Opt-in (stage_backward=True / UIG_DP_STAGED=1): the backward pass of each phase cut into stages (4 for
    the generators, 2 for the discriminators) whose slices of the flat gradient buffer are all-reduced while the next stage runs
    (only the last, smallest bucket - the first layers: 3 MB of 91 MB for the generators - is exposed); UIG_OVERLAP_UPDATE=1 puts
    the generator update (wait for its buckets, Adam, weight repack) on its own stream under the discriminator phase.  Which
    form wins at N > 1 has not been measured on hardware.
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import ops
from .dp import FlatGroup, GradExchange, run_exchange_phase, staged_backward
from .networks import Discriminator, Generator, pair_forward_phys
from .schedule import ImagePool, linear_decay_scale

LOSS_NAMES = ("idt_A", "idt_B", "G_A", "G_B", "cyc_A", "cyc_B", "D_A", "D_B")


class CycleGAN:
    def __init__(self, n_blocks=9, dtype=torch.bfloat16, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8,
                 lambda_cyc=10.0, lambda_idt=0.5, device="cuda", process_group=None, use_graph=False, batch_fused=True, paired=True,
                 force_exchange=False, pool_size=0, pool_seed=0, stage_backward=None, fp8=False):
        """fp8=True (with dtype bf16): mixed-precision step of BASELINE configs[4] - the generators' ResBlock convolutions run
        forward and input gradient on the MX block-scaled fp8 MFMA kernel; everything else, the weight gradients and the
        fp32 master weights / Adam are unchanged."""
        self.device, self.dtype, self.fp8 = torch.device(device), dtype, bool(fp8)
        kw = dict(dtype=dtype, device=device)
        self.G_A, self.G_B = Generator(n_blocks=n_blocks, fp8=fp8, **kw), Generator(n_blocks=n_blocks, fp8=fp8, **kw)
        self.D_A, self.D_B = Discriminator(**kw), Discriminator(**kw)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.lr_scale = 1.0                               # LR-schedule multiplier (set_lr_scale / set_epoch)
        # image history pools feeding the discriminators' fake batch (§8f row 1); size 0 (default, parity / bench) = off
        self.pool_B, self.pool_A = ImagePool(pool_size, pool_seed), ImagePool(pool_size, pool_seed + 1)
        self.lam, self.lam_idt = lambda_cyc, lambda_idt
        self.xchg = GradExchange(process_group, force=force_exchange)
        self.world = self.xchg.world
        self.use_graph, self.batch_fused, self.paired = use_graph, batch_fused, paired
        # parameter-gradient kernels stay on the side stream across layers and are joined once per phase (see ops.deferred_param_grads)
        self.defer_join = os.environ.get("UIG_DEFER_JOIN", "1") != "0"
        # generator update (exchange wait, Adam, repack) on its own stream under the discriminator phase (0: after it, on the main
        # stream).  Default: only when gradients are exchanged (the wait for the all-reduce is what it hides); on one GPU the
        # second stream measured slower (see ops.PARALLEL_BACKWARD)
        # Round 3: OFF by default also under data parallelism (measured at one rank with the exchange forced: 14.39 vs 14.31 ms): the
        # generators' all-reduce is enqueued on the communication stream behind the generator phase and runs under the whole
        # discriminator phase anyway; their Adam + repack then run behind the discriminator phase on the main stream, UNDER the
        # discriminators' all-reduce.
        self.overlap_update = os.environ.get("UIG_OVERLAP_UPDATE", "0") != "0"
        # data-parallel gradient buckets: the backward pass of each phase is cut into stages and each stage's slice of the flat
        # gradient buffer is all-reduced under the next stage (dp.staged_backward / run_exchange_phase).  One stage (= the
        # single-GPU step, bit for bit) when no collective is issued.
        # Round 3: NOT staged by default (UIG_DP_STAGED=1 / stage_backward=True turn it on).  The generators' exchange (91 MB) has the
        # discriminator phase (~3.5 ms) to hide under and the discriminators' (22 MB) the generators' Adam + repack, so cutting the
        # backward passes only buys what it costs: at one rank with the exchange forced the staged step is 14.59 ms, the un-staged one
        # 14.31 ms, no exchange 13.90 ms (one box, bench.py --force-comm).
        self.stage_backward = stage_backward            # None: UIG_DP_STAGED (default 0) when collectives run; True / False: forced
        self.n_stages_G = int(os.environ.get("UIG_DP_STAGES_G", "4"))
        self.n_stages_D = int(os.environ.get("UIG_DP_STAGES_D", "2"))
        self._graphs = None
        if self.device.type == "cuda":
            ops.ticket_arena(self.device)               # arrival-ticket words of the in-launch statistics finalize: before any graph capture
        self._finalize_params()

    # ------------------------------------------------------------------ parameters
    def nets(self):
        return (self.G_A, self.G_B, self.D_A, self.D_B)

    def _finalize_params(self):
        self.grp_G = FlatGroup((self.G_A, self.G_B), self.device)
        self.grp_D = FlatGroup((self.D_A, self.D_B), self.device)
        want = (self.xchg.active and os.environ.get("UIG_DP_STAGED", "0") != "0") if self.stage_backward is None else bool(self.stage_backward)
        staged = want and self.batch_fused and self.paired
        self.cuts_G = self.G_A.stage_cut_modules(self.n_stages_G) if staged else []
        self.cuts_D = self.D_A.stage_cut_modules(self.n_stages_D) if staged else []
        self.buckets_G = self.grp_G.buckets([self.G_A.param_index_at(i) for i in self.cuts_G])
        self.buckets_D = self.grp_D.buckets([self.D_A.param_index_at(i) for i in self.cuts_D])
        self.repack()

    def _stage_params(self, nets, cuts):
        """per backward stage: the parameters (of all `nets`) behind that stage's cut"""
        edges = [0] + [nets[0].param_index_at(i) for i in cuts] + [len(list(nets[0].parameters()))]
        per = [list(n.parameters()) for n in nets]
        return [[p for pl in per for p in pl[a:b]] for a, b in reversed(list(zip(edges, edges[1:])))]

    def load_state_dicts(self, sd_GA, sd_GB, sd_DA, sd_DB):
        """Load torch-layout fp32 weights (e.g. from the stock-torch modules) into the flat buffers."""
        with torch.no_grad():
            for net, sd in zip(self.nets(), (sd_GA, sd_GB, sd_DA, sd_DB)):
                own = dict(net.named_parameters())
                if set(own) != set(sd):
                    raise KeyError(f"state_dict keys differ: {sorted(set(own) ^ set(sd))[:6]}")
                for k, v in sd.items():
                    own[k].copy_(v.to(self.device, torch.float32))
        self.repack()

    def _packer_of(self, which):
        """one-launch weight packer of an optimiser group's networks ('G' or 'D'); rebuilt if buffers were re-allocated"""
        packers = self.__dict__.setdefault("_packers", {})
        mp = packers.get(which)
        if mp is None or not mp.valid():
            nets = (self.G_A, self.G_B) if which == "G" else (self.D_A, self.D_B)
            mp = packers[which] = ops.MultiPacker([l for n in nets for l in n.conv_layers()])
        return mp

    def repack(self):
        """refresh every layer's kernel-side weight operands (packed bf16/fp32 tiles) from the fp32 master copy.  The train
        step keeps them current itself (each Adam is followed by the repack of its group); call this after writing
        parameters from outside."""
        self._packer_of("G").run()
        self._packer_of("D").run()

    def _update_stream(self):
        if not self.overlap_update:
            return torch.cuda.current_stream(self.device)
        st = self.__dict__.get("_upd_stream")
        if st is None:
            st = self._upd_stream = torch.cuda.Stream(device=self.device)
        return st

    def broadcast_params(self, src=0):
        self.xchg.broadcast(self.grp_G.flat, src)
        self.xchg.broadcast(self.grp_D.flat, src)
        if self.world > 1:
            self.repack()

    # ------------------------------------------------------------------ step pieces (all async, static shapes)
    def _g_phase(self, xa, xb):
        """generator forward (6 passes) + D forward (frozen) + 6 losses + backward -> grads in grp_G.grad"""
        res = {}
        for _ in self._g_stages(xa, xb, res):
            pass
        return res["fake_B"], res["fake_A"], res["losses"]

    def _g_stages(self, xa, xb, res):
        """The generator phase as a Python generator over its backward stages (len(self.cuts_G) + 1 of them; one when no
        collective runs): yields the stage index after each stage - the gradients of bucket k = self.buckets_G[k] are then
        complete.  The forward passes and losses run before the first yield.  res receives fake_B, fake_A, losses."""
        B = xa.shape[0]
        taps = {i: None for i in self.cuts_G} if self.cuts_G else None
        self.grp_D.set_requires_grad(False)
        self.grp_G.zero_grad()
        ops.reset_tickets(self.device)                  # one fill per step (every ticketed launch leaves its words zero anyway)
        if self.batch_fused and self.paired:
            # G_A on [xb; xa] and G_B on [xb; xa] as ONE paired pass over 4B images -> [idt_A, fake_B | fake_A, idt_B]: the two
            # fakes are adjacent, so the batch [fake_B; fake_A] that feeds both the cycle pass and the discriminators is a view
            # of the output (no concatenation), and torch.split keeps the backward at ONE concatenation of the three gradient
            # pieces (slicing o four times cost four zero-filled full-size gradients and three adds per step).
            x2 = torch.cat([xb, xa])
            o = pair_forward_phys(self.G_A, self.G_B, torch.cat([x2, x2]), taps)
            idt_A, ff, idt_B = torch.split(o, [B, 2 * B, B])
            fake_B, fake_A = ff[:B], ff[B:]
            r = pair_forward_phys(self.G_B, self.G_A, ff)          # G_B(fake_B) = rec_A ; G_A(fake_A) = rec_B
            rec_A, rec_B = torch.split(r, B)
        elif self.batch_fused:
            oa = self.G_A.forward_phys(torch.cat([xa, xb]))      # [fake_B ; idt_A]
            ob = self.G_B.forward_phys(torch.cat([xb, xa]))      # [fake_A ; idt_B]
            fake_B, idt_A, fake_A, idt_B = oa[:B], oa[B:], ob[:B], ob[B:]
            rec_A, rec_B = self.G_B.forward_phys(fake_B), self.G_A.forward_phys(fake_A)
        else:
            fake_B, fake_A = self.G_A.forward_phys(xa), self.G_B.forward_phys(xb)
            idt_A, idt_B = self.G_A.forward_phys(xb), self.G_B.forward_phys(xa)
            rec_A, rec_B = self.G_B.forward_phys(fake_B), self.G_A.forward_phys(fake_A)
        n_real = B * xa.shape[1] * xa.shape[2] * 3
        l_idt_A = ops.l1_loss(idt_A, xb, self.lam * self.lam_idt, n_real, True)
        l_idt_B = ops.l1_loss(idt_B, xa, self.lam * self.lam_idt, n_real, True)
        if self.batch_fused and self.paired:
            pd_A, pd_B = torch.split(pair_forward_phys(self.D_A, self.D_B, ff), B)
            l_G_A, l_G_B = ops.mse_const(pd_A, 1.0, 1.0, True), ops.mse_const(pd_B, 1.0, 1.0, True)
        else:
            l_G_A = ops.mse_const(self.D_A.forward_phys(fake_B), 1.0, 1.0, True)
            l_G_B = ops.mse_const(self.D_B.forward_phys(fake_A), 1.0, 1.0, True)
        l_cyc_A = ops.l1_loss(rec_A, xa, self.lam, n_real, True)
        l_cyc_B = ops.l1_loss(rec_B, xb, self.lam, n_real, True)
        losses = [l_idt_A, l_idt_B, l_G_A, l_G_B, l_cyc_A, l_cyc_B]
        res.update(fake_B=fake_B.detach(), fake_A=fake_A.detach(), losses=losses)
        self.last_fake_B = res["fake_B"]
        ctx = (lambda k: ops.deferred_param_grads(self.device)) if self.defer_join else None
        # both generator passes back-propagate through the same layer pairs: one weight-gradient launch per pair for both batches.
        # The LAST stage index is yielded only after the region has closed (its exit flushes a layer pair that was visited once
        # into grp_G.grad) and the discriminators are un-frozen: whoever drives this generator starts the last bucket's all-reduce
        # at that yield, so nothing may write a gradient behind it - in eager mode as in a captured stage graph.
        last = None
        with ops.combined_pass_wgrad(self.device) if (self.batch_fused and self.paired) else contextlib.nullcontext():
            if taps:
                # pass 1 (the 4B-image pass) is the LAST part of the backward pass and a chain through its ResBlocks: cut there
                cuts = [taps[i] for i in self.cuts_G]
                n_st = len(cuts) + 1
                for k in staged_backward(losses, cuts, self._stage_params((self.G_A, self.G_B), self.cuts_G), ctx):
                    # stages 0 .. n-2 are yielded the moment they finish (their buckets lie behind a cut of pass 1: every layer pair in
                    # them has been visited by both passes, nothing is left in the stash for them) so that bucket k's all-reduce runs
                    # under stage k+1; only the final index waits for the region's exit flush
                    if k < n_st - 1:
                        yield k
                    else:
                        last = k
            else:
                with ctx(0) if ctx else contextlib.nullcontext():
                    ops.backward_unit(losses)
                last = 0
        self.grp_D.set_requires_grad(True)
        yield last

    def _d_phase(self, xa, xb, fake_B, fake_A):
        res = {}
        for _ in self._d_stages(xa, xb, fake_B, fake_A, res):
            pass
        return res["losses"]

    def _d_stages(self, xa, xb, fake_B, fake_A, res):
        """The discriminator phase as a generator over its backward stages (see _g_stages); res["losses"] = [(real, fake) x 2]."""
        self.grp_D.zero_grad()
        B = xa.shape[0]
        out = []
        if self.batch_fused and self.paired:      # D_A on [real_B; fake_B] and D_B on [real_A; fake_A] as one paired pass
            taps = {i: None for i in self.cuts_D} if self.cuts_D else None
            p = torch.split(pair_forward_phys(self.D_A, self.D_B, torch.cat([xb, fake_B, xa, fake_A]), taps), B)
            ls = [ops.mse_const(p[0], 1.0, 0.5, True), ops.mse_const(p[1], 0.0, 0.5, True),
                  ops.mse_const(p[2], 1.0, 0.5, True), ops.mse_const(p[3], 0.0, 0.5, True)]
            res["losses"] = [(ls[0], ls[1]), (ls[2], ls[3])]
            ctx = (lambda k: ops.deferred_param_grads(self.device)) if self.defer_join else None
            if taps:
                yield from staged_backward(ls, [taps[i] for i in self.cuts_D], self._stage_params((self.D_A, self.D_B), self.cuts_D), ctx)
            else:
                with ctx(0) if ctx else contextlib.nullcontext():
                    ops.backward_unit(ls)
                yield 0
            return
        for D, real, fake in ((self.D_A, xb, fake_B), (self.D_B, xa, fake_A)):
            if self.batch_fused:
                p = D.forward_phys(torch.cat([real, fake]))
                l_real, l_fake = ops.mse_const(p[:B], 1.0, 0.5, True), ops.mse_const(p[B:], 0.0, 0.5, True)
            else:
                l_real, l_fake = ops.mse_const(D.forward_phys(real), 1.0, 0.5, True), ops.mse_const(D.forward_phys(fake), 0.0, 0.5, True)
            ops.backward_unit([l_real, l_fake])
            out.append((l_real, l_fake))
        res["losses"] = out
        yield 0

    def _adam(self, grp):
        grp.step += 1
        ops.adam_flat(grp.flat, grp.grad, grp.m, grp.v, self.lr * self.lr_scale, self.b1, self.b2, self.eps, grp.step, 1.0 / self.world)

    # ------------------------------------------------------------------ schedule / pool / checkpoint (§8f rows 1-2)
    def set_lr_scale(self, scale: float):
        """LR-schedule multiplier for the following steps (both optimisers).  In graph mode it is written into the Adam
        device records, so the captured graphs keep replaying - no re-capture."""
        self.lr_scale = float(scale)
        for grp in (self.grp_G, self.grp_D):
            st = getattr(grp, "state16", None)
            if st is not None:
                st.view(torch.float32)[3:4].fill_(self.lr_scale)

    def set_epoch(self, epoch: int, n_const: int = 100, n_decay: int = 100):
        """constant LR for n_const epochs, then linear decay to zero over n_decay epochs [PAPER]"""
        self.set_lr_scale(linear_decay_scale(epoch, n_const, n_decay))

    def _pool_fakes(self, fake_B, fake_A, out_B=None, out_A=None):
        return self.pool_B.query(fake_B, out_B), self.pool_A.query(fake_A, out_A)

    def state_dict(self):
        """Everything needed to resume: the four networks (stock-torch state_dict keys), both Adam states, schedule, pools."""
        torch.cuda.synchronize(self.device)
        sd = {"nets": [{k: v.detach().clone() for k, v in n.state_dict().items()} for n in self.nets()],
              "lr_scale": self.lr_scale, "pool_B": self.pool_B.state_dict(), "pool_A": self.pool_A.state_dict()}
        for name, grp in (("G", self.grp_G), ("D", self.grp_D)):
            step = grp.step
            st = getattr(grp, "state16", None)
            if st is not None and self.graph_active:        # graph mode keeps the authoritative counter on the device
                step = int(st[0].item())
            # Adam moments PER PARAMETER, keyed by network and state_dict name: independent of how FlatGroup lays the flat buffers
            # out (the interleaved layout of round 2 permuted parameters at unchanged numel - a raw flat copy would load silently wrong)
            keys = self._param_keys(name)
            sd["opt_" + name] = {"format": "per_param_v1", "step": step,
                                 "m": {k: t.clone() for k, t in zip(keys, grp.param_views(grp.m))},
                                 "v": {k: t.clone() for k, t in zip(keys, grp.param_views(grp.v))}}
        return sd

    def _param_keys(self, which):
        """'<net>.<state_dict name>' of every parameter of an optimiser group, in the order of its FlatGroup.params"""
        nets = (("G_A", self.G_A), ("G_B", self.G_B)) if which == "G" else (("D_A", self.D_A), ("D_B", self.D_B))
        grp = self.grp_G if which == "G" else self.grp_D
        by_id = {id(p): f"{nn}.{k}" for nn, net in nets for k, p in net.named_parameters()}
        return [by_id[id(p)] for p in grp.params]

    def load_state_dict(self, sd):
        self.load_state_dicts(*sd["nets"])
        for name, grp in (("G", self.grp_G), ("D", self.grp_D)):
            o = sd["opt_" + name]
            if o.get("format") != "per_param_v1":
                raise ValueError("checkpoint holds Adam state as raw flat buffers (written before the per-parameter format): their "
                                 "parameter order is not recorded and differs between layouts of the same size - refusing to load it")
            keys = self._param_keys(name)
            if set(keys) != set(o["m"]) or set(keys) != set(o["v"]):
                raise KeyError(f"optimizer state keys differ: {sorted(set(keys) ^ set(o['m']))[:6]}")
            for k, mv, vv in zip(keys, grp.param_views(grp.m), grp.param_views(grp.v)):
                mv.copy_(o["m"][k]); vv.copy_(o["v"][k])
            grp.step = int(o["step"])
            st = getattr(grp, "state16", None)
            if st is not None:
                st[0] = grp.step
        self.pool_B.load_state_dict(sd["pool_B"]); self.pool_A.load_state_dict(sd["pool_A"])
        self.set_lr_scale(sd.get("lr_scale", 1.0))

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))      # tensors, ints, floats, tuples only

    # ------------------------------------------------------------------ the step
    def _step_eager(self, xa, xb):
        """Stream plan: the generator update (gradient exchange, Adam, repack of the generators' kernel operands) does not
        depend on the discriminator phase and the discriminator phase does not read the generators' weights, so the update
        runs on its own stream UNDER the discriminators' forward+backward (it was ~0.45 ms of the critical path)."""
        main, upd = torch.cuda.current_stream(self.device), self._update_stream()
        rg, rd = {}, {}
        # bucket k's all-reduce starts the moment backward stage k has produced it and runs under the stages that follow
        h_g = run_exchange_phase(self._g_stages(xa, xb, rg), self.xchg, self.grp_G.grad, self.buckets_G)
        fake_B, fake_A, lg = rg["fake_B"], rg["fake_A"], rg["losses"]
        if self.overlap_update:
            upd.wait_stream(main)
            with torch.cuda.stream(upd):
                self.xchg.wait_all(h_g, self.device)
                self._adam(self.grp_G)
                self._packer_of("G").run()
        fake_B, fake_A = self._pool_fakes(fake_B, fake_A)
        h_d = run_exchange_phase(self._d_stages(xa, xb, fake_B, fake_A, rd), self.xchg, self.grp_D.grad, self.buckets_D)
        ld = rd["losses"]
        if not self.overlap_update:
            self.xchg.wait_all(h_g, self.device)
            self._adam(self.grp_G)
            self._packer_of("G").run()
        self.xchg.wait_all(h_d, self.device)
        self._adam(self.grp_D)
        self._packer_of("D").run()
        main.wait_stream(upd)
        l_D_A = ld[0][0] + ld[0][1]
        l_D_B = ld[1][0] + ld[1][1]
        return torch.cat(lg + [l_D_A, l_D_B])

    @property
    def graph_active(self) -> bool:
        """True while train_step really replays captured HIP graphs (False before the first step, in eager mode, and after a
        refused capture made graph_train_step fall back to eager launches)."""
        return bool(self.use_graph and self._graphs is not None)

    def close(self):
        """Ordered teardown of everything this model holds on the device BEFORE the caller destroys the process group or the
        interpreter exits: drain the device, then drop the captured graphs (their executables and private memory pool), the
        weight packers' descriptor tables and the side streams, and drain again.  Left to interpreter shutdown these are
        finalized in arbitrary order relative to RCCL's communicator and torch's allocator, which round 1 saw abort at exit.
        The model is unusable for graph replay afterwards (a later train_step would capture again)."""
        import gc
        torch.cuda.synchronize(self.device)
        st, self._graphs = self._graphs, None
        if st is not None:
            for name in ("g4", "g3", "g2", "g1"):
                if hasattr(st, name):
                    delattr(st, name)                       # g1 / g2 are lists of stage graphs
            st.__dict__.clear()
        del st
        # tensors that live in the graphs' private memory pool (last outputs) and the device-side Adam records go with the graphs:
        # nothing of this model may keep a block of that pool, or a pointer a later capture could bake in, alive
        for name in ("_packers", "_upd_stream", "last_fake_B", "last_losses"):
            self.__dict__.pop(name, None)
        for grp in (self.grp_G, self.grp_D):
            grp.__dict__.pop("state16", None)           # graph_train_step keeps grp.step in step with the device counter
        self.xchg.close()
        ops.release_side_streams(self.device)
        gc.collect()
        torch.cuda.synchronize(self.device)

    def to_phys(self, x):
        """logical (B,3,H,W) -> physical (B,H,W,8) in the compute dtype; a tensor that already is physical (as the
        device-side input pipeline produces it) passes through untouched."""
        if x.dim() == 4 and x.shape[3] == 8 and x.shape[1] != 3 and x.dtype == self.dtype:
            return x
        return ops.to_nhwc(x, self.dtype)

    def train_step(self, real_A: torch.Tensor, real_B: torch.Tensor, sync: bool = True):
        """One optimisation step. real_*: logical (B,3,H,W), or physical (B,H,W,8) batches from pipeline.py. Returns the 8 losses (dict of floats, or a device tensor
        of shape (8,) in LOSS_NAMES order when sync=False)."""
        if self.use_graph:
            from .graph_step import graph_train_step
            losses = graph_train_step(self, real_A, real_B)
        else:
            losses = self._step_eager(self.to_phys(real_A), self.to_phys(real_B))
        self.last_losses = losses
        if not sync:
            return losses
        ops.check_sync_errors(self.device)              # the fused kernels' bounded in-kernel waits (one device word; the host synchronises here anyway)
        return dict(zip(LOSS_NAMES, self.xchg.mean_scalars(losses).tolist()))
