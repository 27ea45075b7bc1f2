"""Data-parallel plumbing: flat parameter/gradient buffers and the gradient exchange.

One process per GPU; replicas hold identical parameters; InstanceNorm statistics are per sample, so the only exchange
is a sum-all-reduce of each optimiser group's flat fp32 gradient buffer per step (RCCL over xGMI when the process group's
backend is "nccl"; gloo on CPU in the tests).  Default (round 3): the whole buffer as ONE bucket, started on the communication
stream right behind its phase (it runs under the next phase).  Staged form (opt-in: CycleGAN(stage_backward=True) /
UIG_DP_STAGED=1): issued in BUCKETS - the backward pass is cut into stages (staged_backward), the parameters behind each cut are
one contiguous slice of the flat buffer (FlatGroup's interleaved layout), and a slice's all-reduce is started as soon as its
stage has finished, so it runs under the next stage's kernels (run_exchange_phase).  The 1/world_size average is folded into
the Adam kernel.
This module has no dependency on the HIP library, so its logic is covered by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGroup:
    """All parameters of an optimiser group as views into one flat fp32 buffer (+ flat grad / Adam m, v).

    Layout: networks of identical architecture (G_A / G_B, D_A / D_B) are interleaved parameter by parameter
    (A.p0, B.p0, A.p1, B.p1, ...), so the parameters of a RANGE OF LAYERS of all networks are one contiguous slice: the unit
    of the bucketed gradient exchange (bucket_range).  Backward produces gradients from the last layer to the first, so the
    slice of the last layers is complete first."""

    def __init__(self, nets, device):
        per_net = [list(n.parameters()) for n in nets]
        same = all(len(pl) == len(per_net[0]) and all(a.shape == b.shape for a, b in zip(pl, per_net[0])) for pl in per_net)
        self.interleaved = same and len(per_net) > 1
        if self.interleaved:
            self.params = [pl[i] for i in range(len(per_net[0])) for pl in per_net]
        else:
            self.params = [p for pl in per_net for p in pl]
        self.n_nets, self.params_per_net = len(per_net), len(per_net[0])
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]        # keep every view 16-byte aligned
        total = sum(sizes)
        self._starts = [0]
        for sz in sizes:
            self._starts.append(self._starts[-1] + sz)
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.grad = torch.zeros(total, device=device, dtype=torch.float32)
        self.m = torch.zeros(total, device=device, dtype=torch.float32)
        self.v = torch.zeros(total, device=device, dtype=torch.float32)
        off = 0
        for p, sz in zip(self.params, sizes):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            off += sz
        self.step = 0

    def zero_grad(self):
        self.grad.zero_()

    def param_views(self, flat: torch.Tensor):
        """views of a flat buffer of this group's layout (grad / m / v), one per parameter, in self.params order and shape"""
        return [flat[a:a + p.numel()].view(p.shape) for a, p in zip(self._starts, self.params)]

    def bucket_range(self, p0: int, p1: int):
        """(start, end) of the flat slice holding parameters p0 <= index < p1 (index within ONE network's parameters()
        order) of every network of the group.  Needs the interleaved layout (or a single network)."""
        if not (self.interleaved or self.n_nets == 1):
            raise ValueError("bucket_range needs networks of identical architecture")
        k = self.n_nets
        return self._starts[p0 * k], self._starts[p1 * k]

    def buckets(self, cuts):
        """Flat slices in BACKWARD order for ascending parameter-index cut points: cuts [c1 < c2 < ...] give the buckets
        [c_last, n), ..., [c1, c2), [0, c1): bucket k is complete after backward stage k (staged_backward)."""
        edges = [0] + list(cuts) + [self.params_per_net]
        if any(b <= a for a, b in zip(edges, edges[1:])):
            raise ValueError(f"bucket cut points must be strictly increasing inside (0, {self.params_per_net}): {list(cuts)}")
        return [self.bucket_range(a, b) for a, b in reversed(list(zip(edges, edges[1:])))]

    def set_requires_grad(self, flag: bool):
        for p in self.params:
            p.requires_grad_(flag)


def staged_backward(roots, cut_tensors, stage_params=None, enter_stage=None):
    """Generator: the backward pass of `roots` (each with gradient 1) cut into len(cut_tensors) + 1 stages.

    cut_tensors: activations of a CHAIN through which every path from the roots to the earlier layers passes, in FORWARD
    order (t1 before t2 before ...); an entry may be a list of tensors when several independent chains run side by side
    (two networks evaluated separately): together they must cut every path.  Stage 0 back-propagates from the roots down to the last cut tensor, stage k from cut
    tensor K-k+1 down to cut tensor K-k, the last stage from the first cut tensor to the leaves.  After each stage the
    generator yields its index: at that point the gradients of every parameter BEHIND that stage's cut are final, so the
    caller can start that bucket's all-reduce while the next stage runs (run_exchange_phase).
    stage_params[k]: parameters whose .grad stage k must produce (needed for stock autograd ops, which only compute what
    `inputs=` names; the HIP operators accumulate in place regardless).  enter_stage(k): optional context manager per stage
    (the HIP path joins its parameter-gradient side stream on exit)."""
    import contextlib
    ones = torch.ones((1,), device=roots[0].device, dtype=roots[0].dtype)
    cur, grads = list(roots), [ones.reshape(r.shape) if r.numel() == 1 else torch.ones_like(r) for r in roots]
    cuts = list(cut_tensors)
    n = len(cuts) + 1
    for k in range(n):
        last = k == n - 1
        with (enter_stage(k) if enter_stage is not None else contextlib.nullcontext()):
            if last:
                torch.autograd.backward(cur, grads)
            else:
                c = cuts[n - 2 - k]
                cs = list(c) if isinstance(c, (list, tuple)) else [c]
                extra = list(stage_params[k]) if stage_params is not None else []
                # retain_graph: the stages are disjoint parts of ONE graph; without it the engine frees saved tensors of
                # nodes the later stages still have to run
                torch.autograd.backward(cur, grads, inputs=cs + extra, retain_graph=True)
                cur, grads = cs, [t.grad for t in cs]
        yield k


def run_exchange_phase(stages, xchg, flat_grad, buckets):
    """Drive one phase's backward stages and start bucket k's all-reduce the moment stage k has produced it (asynchronous:
    it runs on the communication stream under stage k+1).  Returns the handles, in bucket order, for wait_all."""
    handles = []
    for k in stages:
        a, b = buckets[k]
        handles.append(xchg.start(flat_grad[a:b]))
    if len(handles) != len(buckets):
        raise RuntimeError(f"{len(handles)} backward stages for {len(buckets)} gradient buckets")
    return handles


class GradExchange:
    """Asynchronous sum-all-reduce of flat gradient buffers.  start() returns a handle; wait() makes the CURRENT stream
    (or, on CPU, the caller) wait for it.  world_size 1 is a no-op, so the single-GPU step is bit-identical with or
    without a process group."""

    def __init__(self, process_group=None, force: bool = False):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.force = force and dist.is_available() and dist.is_initialized()   # exercise the collective even at world 1
        self._stream = None
        self.n_started = 0                                  # collectives really issued (tests assert the exchange path ran)

    def close(self):
        """drop the communication stream (after the device has been drained by the owner)"""
        self._stream = None

    def start(self, flat: torch.Tensor):
        if self.world <= 1 and not self.force:
            return None
        self.n_started += 1
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))   # gradients are complete on the main stream
            with torch.cuda.stream(self._stream):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            ev = torch.cuda.Event()
            ev.record(self._stream)
            return ev
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def wait(self, handle, device=None):
        if handle is None:
            return
        if isinstance(handle, torch.cuda.Event):
            torch.cuda.current_stream(device).wait_event(handle)
        else:
            handle.wait()

    def wait_all(self, handles, device=None):
        for h in handles:
            self.wait(h, device)

    @property
    def active(self) -> bool:
        """True when start() really issues collectives (world > 1, or forced for plumbing tests)"""
        return self.world > 1 or self.force

    def mean_scalars(self, t: torch.Tensor) -> torch.Tensor:
        """average a small tensor of logged scalars over the ranks (logging only; not on the data path)"""
        if self.world <= 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t / self.world

    def broadcast(self, flat: torch.Tensor, src: int = 0):
        if self.world > 1:
            dist.broadcast(flat, src, group=self.pg)
