"""Data-parallel plumbing: flat parameter/gradient buffers and the gradient exchange.

One process per GPU; replicas hold identical parameters; InstanceNorm statistics are per sample, so the only exchange
is ONE sum-all-reduce per optimiser group per step over its flat fp32 gradient buffer (RCCL over xGMI when the process
group's backend is "nccl"; gloo on CPU in the tests).  The 1/world_size average is folded into the Adam kernel.
The all-reduce is enqueued on a dedicated communication stream so that it overlaps the compute that follows on the main
stream (generator grads <-> discriminator forward+backward; discriminator grads <-> generator Adam).
This module has no dependency on the HIP library, so its logic is covered by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGroup:
    """All parameters of an optimiser group as views into one flat fp32 buffer (+ flat grad / Adam m, v)."""

    def __init__(self, nets, device):
        self.params = [p for n in nets for p in n.parameters()]
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]        # keep every view 16-byte aligned
        total = sum(sizes)
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

    def set_requires_grad(self, flag: bool):
        for p in self.params:
            p.requires_grad_(flag)


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
