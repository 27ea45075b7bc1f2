"""Operator boundary (layer L2): one torch.autograd.Function per op, each a thin host wrapper that
enqueues hand-written HIP kernels through the C ABI (include/uig.h) on torch's current stream.

Tensors between ops are PHYSICAL NHWC: shape (B, H, W, Cp), contiguous, channels padded to a multiple of 8 with
zeros (a 1-channel discriminator output stays unpadded).  torch is used for device memory (caching allocator),
streams and autograd bookkeeping only; every FLOP on these paths runs in libuig.so.  Missing library => RuntimeError.
"""
from __future__ import annotations

import contextlib
import os
import threading

import torch
from torch.autograd import Function

from . import lib as L


REFLECT_DGRAD_DIRECT = os.environ.get("UIG_REFLECT_DGRAD_DIRECT", "1") != "0"   # 3x3 reflect-pad convs: input gradient on the exact grid + border GEMM (no padded gradient, no fold)
PAIR_WGRAD = os.environ.get("UIG_PAIR_WGRAD", "1") != "0"                   # paired layers: both networks' weight-gradient partials in one launch where the library supports it
FUSE_SKIP_GRAD = os.environ.get("UIG_FUSE_SKIP_GRAD", "1") != "0"           # ResBlock: the skip path's gradient is added in conv1's input-gradient epilogue instead of by a separate add kernel
# Run a conv's parameter-gradient kernels on a side stream beside its input-gradient kernel.  OFF by default since round 2: the
# MFMA kernels of the backward pass (strip dgrad: 149.5 KB LDS, 246 VGPRs x 2 waves per SIMD; image-row wgrad: 132 KB, same
# registers) each fill a CU on their own, so two of them never share one and the HBM-bound InstanceNorm kernels find no free
# register file beside them either - the second stream only adds contention: 14.01 vs 14.25 ms per step (two boxes, A/B in one call).
PARALLEL_BACKWARD = os.environ.get("UIG_PARALLEL_BACKWARD", "0") != "0"
# InstanceNorm backward statistics (sum g, sum g*xhat) from the epilogue of the input-gradient launch that writes the norm's dy,
# instead of the norm's own pass over dy and x.  OFF by default: measured on MI355X (scripts/bench_dgrad_nd.py, paired 16-image
# launch) the epilogue work costs +13.7 us against the 17 us statistics kernel it removes (+9.2 vs 9 us at 8 images); whole step
# 16.08-16.33 ms with it vs 15.98-16.00 ms without (same box).  ~700 VALU instructions per wave at 2 waves per SIMD sit on the
# convolution's critical path, while the stand-alone pass is HBM-bound at full occupancy.  Tested opt-in (test_ops_gpu.py).
FUSE_BWD_STATS = os.environ.get("UIG_FUSE_BWD_STATS", "0") != "0"
MX_DGRAD_MIRROR = os.environ.get("UIG_MX_DGRAD_MIRROR", "1") != "0"   # fp8 layers: reflect-pad input gradient in one launch (mirror pixels re-quantised in LDS) instead of fp8 main term + bf16 border GEMM
FUSE_MX_QUANT = os.environ.get("UIG_FUSE_MX_QUANT", "1") != "0"            # fp8 path: MX quantisation of activations / gradients inside the InstanceNorm launches
# ResBlock: conv2 applies the InstanceNorm + ReLU in front of it to its own input strip (NormConvFn): no apply pass between the block's two convolutions
NORM_CONV = os.environ.get("UIG_NORM_CONV", "0") != "0"      # OFF by default: measured slower (13.64 vs 13.48 ms per step; g_fwd 2.45 vs 2.34 ms) - see DESIGN.md
COMBINE_PASS_WGRAD = os.environ.get("UIG_COMBINE_PASS_WGRAD", "1") != "0"  # one weight-gradient launch per ResBlock conv pair for BOTH generator passes of a step
# Round 4: InstanceNorm statistics are finalised INSIDE the launch that produces their partial slabs (arrival tickets: the image's
# last-arriving block reduces the slabs; include/uig.h, uig_conv_gather_fin) instead of by a finalize launch (112 per train step,
# 5.3 us each).  0 = the finalize launches (bit-identical results: A/B and the parity test).
IN_TICKETS = os.environ.get("UIG_IN_TICKETS", "0") != "0"
# Round 4: the mirror-pixel input-gradient launch of a ResBlock convolution also emits the statistics of the InstanceNorm backward that
# consumes its output (and finalises them): that norm's own statistics pass (a full read of dy and x) and its finalize launch are gone.
MIRROR_BST = os.environ.get("UIG_MIRROR_BST", "0") != "0"
# Round 4: the InstanceNorm backward as ONE launch and one pass over dy and x (statistics + finalize + apply fused, the blocks of an image
# synchronise inside the kernel: include/uig.h, uig_instnorm_act_bwd_fused) wherever the whole grid can be resident at once (the ResBlock
# maps of the 256x256 configurations, the PatchGAN's norms).  OPT-IN: measured slower on MI355X (scripts/bench_in_fused.py, 64x64x256
# maps: 8 images 34.6 us vs 27.9 us for the three launches; 16 images 99 us vs 37 - the 1024-block grid equals the nominal residency
# limit and does not become resident at once; whole step 15.29 vs 13.49 ms).  Each of the two in-kernel synchronisations costs ~8 us
# (write-through drain, counter add, poll, sc1 re-read: four memory round trips of ~2 us), more than the two launch boundaries and the
# second read of dy and x (from the 256-MB Infinity Cache) they replace.
FUSED_IN_BWD = os.environ.get("UIG_FUSED_IN_BWD", "0") != "0"
_TICKET_WORDS = 4096     # images per ticketed launch (one 32-bit word each) and family
_TICKETS = {}            # device index -> int32 arena [3 * _TICKET_WORDS + 16], zero between launches: family 0 = forward statistics,
                         # 1 = backward statistics, 2 = in-kernel synchronisation words of the fused backward (4 per image); the word at
                         # 3 * _TICKET_WORDS is the fused kernels' error flag (a bounded in-kernel wait ran out): check_sync_errors


def _dev_index(device) -> int:
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def ticket_arena(device) -> torch.Tensor:
    """The device's arrival-ticket words (all zero between launches: the last-arriving block of a ticketed launch resets its word).
    Launches that use them are ordered by the stream they run on; the two families (forward / backward statistics) use disjoint halves.
    Allocated on first use - which must not be inside a graph capture (the words would live in the graph's private pool): models
    that capture call this in their constructor."""
    idx = _dev_index(device)
    t = _TICKETS.get(idx)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops.ticket_arena(device) must be called once before the first graph capture")
        t = _TICKETS[idx] = torch.zeros(3 * _TICKET_WORDS + 16, device=torch.device("cuda", idx), dtype=torch.int32)
    return t


def _tickets(device, family: int, B: int):
    """pointer to the ticket words of `family` for a launch over B images, or None (in-launch finalize off / batch too large)"""
    if not IN_TICKETS or B > _TICKET_WORDS:
        return None
    return ticket_arena(device).data_ptr() + family * _TICKET_WORDS * 4


def reset_tickets(device) -> None:
    """zero the arena (one fill per train step, ahead of the first ticketed launch): a launch that was aborted mid-way cannot poison
    the following steps"""
    if IN_TICKETS or FUSED_IN_BWD:
        ticket_arena(device)[:3 * _TICKET_WORDS].zero_()      # not the error flag


def check_sync_errors(device) -> None:
    """Raise if a fused kernel's bounded in-kernel wait ran out since the last check (its results were then garbage): that can only
    happen when the kernel's grid was not fully resident - another stream's kernels holding compute units for longer than the ~1 s
    bound, or fewer units than the occupancy query promised.  Reads one device word (a host synchronisation: call it where the host
    synchronises anyway).  After an error the fused path is switched off for the rest of the process."""
    global FUSED_IN_BWD
    t = _TICKETS.get(_dev_index(device))
    if t is None or not FUSED_IN_BWD:
        return
    if int(t[3 * _TICKET_WORDS].item()) != 0:
        t[3 * _TICKET_WORDS:].zero_()
        FUSED_IN_BWD = False
        raise RuntimeError("uig: an in-kernel wait of the fused InstanceNorm backward timed out (grid not co-resident); its results were invalid. "
                           "The fused path is now disabled for this process (UIG_FUSED_IN_BWD=0 selects the three-launch form from the start).")

_SIDE_STREAMS = {}
_DEFER_JOIN = {}
_WG_STASH = {}          # device index -> {layer pair: (x, dy, group, layers)} while a combined_pass_wgrad region is active


def _side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def release_side_streams(device) -> None:
    """forget everything this module holds for `device` between calls (CycleGAN.close(): ordered teardown): the
    parameter-gradient side stream, the deferred-join flag and any weight-gradient stash an interrupted backward left behind
    (it would pin activations of a dead model and be picked up by the next model's first layer pair)"""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    _SIDE_STREAMS.pop(idx, None)
    _DEFER_JOIN.pop(idx, None)
    _WG_STASH.pop(idx, None)
    _TICKETS.pop(idx, None)          # re-created (outside any capture) by the next model's constructor


def device_state_empty() -> bool:
    """True when no stream, flag or stashed tensor of any model is held at module level (after every model was closed)"""
    return not _SIDE_STREAMS and not _WG_STASH and not any(_DEFER_JOIN.values()) and not _TICKETS


class combined_pass_wgrad:
    """Region (the generator phase's backward pass) in which a layer pair that is back-propagated TWICE - the step's two generator
    passes run the same two weight sets, the second pass's backward first - gets ONE weight-gradient launch for both batches:
    the first visit only stashes its (x, dy, column-sum partials for the bias gradient), the second visit runs uig_wgrad_partial_pair2 over
    both batches and the usual paired reduce.  The fixed cost of a split-K launch (fill / drain, partial slabs, reduce) is paid
    once per layer instead of twice: 147 us against 111 + 74 us per ResBlock conv pair at batch 4 (scripts/bench_wgrad_combine.py).
    Anything still stashed on exit (a pair visited once) is flushed with an ordinary launch."""

    def __init__(self, device):
        self.idx = torch.device(device).index
        if self.idx is None:
            self.idx = torch.cuda.current_device()

    def __enter__(self):
        if COMBINE_PASS_WGRAD:
            _WG_STASH[self.idx] = {}
        return self

    def __exit__(self, *exc):
        st = _WG_STASH.pop(self.idx, None)
        if st:
            side = _side_stream(torch.device("cuda", self.idx))
            main = torch.cuda.current_stream(self.idx)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for x, dy, group, layers, cs in st.values():
                    _stashed_alone(x, dy, group, layers, cs)
            main.wait_stream(side)
        return False


def _combined_wgrad(layers, spec, x, dy, group, colsum, stash):
    """weight (+ bias) gradients of a layer pair inside a combined_pass_wgrad region; returns False if the ordinary path must run"""
    lib = L.lib()
    key = frozenset((id(layers[0]), id(layers[1])))
    Pt, Qt, Mh, Mw, Np, Hq, Wq, Cq, pm, D0, D1 = _wgrad_operands(spec, x, dy)
    B = x.shape[0]
    prev = stash.pop(key, None)
    if prev is None:
        if int(lib.uig_wgrad_pair2_splits(B, group, B, group, 0, Mh, Mw, Np, Hq, Wq, Cq, spec.k, spec.k, spec.stride, spec.pad, _dt(x))) <= 0:
            return False
        # first visit (the later pass): weight AND bias gradient together with the other pass's batch (its column-sum partials
        # travel with the stash and ride on the combined reduce); no partials -> the bias gradient now
        cs = colsum if (colsum is not None and colsum[2] == dy.shape[3]) else None
        if cs is None:
            for i, l in enumerate(layers):
                bias_grad(dy[:group] if i == 0 else dy[group:], spec.cout, out=l.bias.grad, accumulate=True)
        stash[key] = (x, dy, group, layers, cs)
        return True
    x2, dy2, g2, layers2, cs2 = prev
    swap2 = 1 if layers2[0] is layers[1] else 0
    P2, Q2 = _wgrad_operands(spec, x2, dy2)[:2]
    B2 = x2.shape[0]
    splits = int(lib.uig_wgrad_pair2_splits(B, group, B2, g2, swap2, Mh, Mw, Np, Hq, Wq, Cq, spec.k, spec.k, spec.stride, spec.pad, _dt(x)))
    ok = splits > 0 and tuple(x2.shape[1:]) == tuple(x.shape[1:]) and (layers2[0] is layers[swap2]) and (layers2[1] is layers[1 - swap2])
    if ok:
        per = splits * Np * spec.k * spec.k * Cq
        ws = torch.empty((2 * per,), device=x.device, dtype=torch.float32)
        L.check(lib.uig_wgrad_partial_pair2(_p(Pt), _p(Qt), _p(P2), _p(Q2), _p(ws), B, group, B2, g2, swap2, Mh, Mw, Np, Hq, Wq, Cq,
                                            spec.k, spec.k, spec.stride, spec.pad, pm, splits, _dt(x), _stream()), "uig_wgrad_partial_pair2")
        c2 = None
        if cs2 is not None:      # images of network a (= layers[0]) / b in the stashed batch
            c2 = (cs2, (g2, B2 - g2), (0, g2)) if swap2 else (cs2, (0, g2), (g2, B2 - g2))
            if not (colsum is not None and colsum[2] == dy.shape[3]):      # no rider on this pass to carry them
                for (i0, n), l in zip(c2[1:], layers):
                    _bias_grad_from_partials(cs2, i0, n, spec.cout, l.bias.grad, True)
                c2 = cs2 = None
        if _param_grads_pair(layers, spec, x, dy, group, colsum, [(ws[:per], splits), (ws[per:], splits)], colsum2=c2):
            return True
    # could not combine after all: the stashed batch on its own, then the ordinary path for this one
    _stashed_alone(x2, dy2, g2, layers2, cs2)
    return False


def _stashed_alone(x, dy, group, layers, cs):
    """weight (and, if its column-sum partials `cs` are still pending, bias) gradients of a stashed batch by itself"""
    spec = layers[0].spec
    pp = conv_wgrad_pair_partial(spec, x, dy, group)
    if pp and _param_grads_pair(layers, spec, x, dy, group, cs, pp, bias=cs is not None):
        return
    for i, l in enumerate(layers):
        i0, n = (0, group) if i == 0 else (group, x.shape[0] - group)
        conv_wgrad(spec, x[i0:i0 + n], dy[i0:i0 + n], out=l.weight.grad, accumulate=True, partial=pp[i] if pp else None)
        if cs is not None:
            _bias_grad_from_partials(cs, i0, n, spec.cout, l.bias.grad, True)


class deferred_param_grads:
    """Context manager for a backward region whose parameter gradients are accumulated in place (trainer-owned flat
    gradient buffers): inside it the conv backward does not join its side stream after every layer; on exit the main
    stream waits for the side stream once.  Nothing inside the region may read the .grad buffers."""

    def __init__(self, device):
        self.idx = torch.device(device).index
        if self.idx is None:
            self.idx = torch.cuda.current_device()

    def __enter__(self):
        _DEFER_JOIN[self.idx] = True
        return self

    def __exit__(self, *exc):
        _DEFER_JOIN[self.idx] = False
        st = _SIDE_STREAMS.get(self.idx)
        if st is not None:
            torch.cuda.current_stream(self.idx).wait_stream(st)
        return False


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk_phys(t: torch.Tensor, name: str) -> None:
    if not (t.is_cuda and t.dim() == 4 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous CUDA tensor (B,H,W,C), got {tuple(t.shape)} on {t.device}")


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ----------------------------------------------------------------------------------------- layout plumbing
def to_nhwc(x: torch.Tensor, dtype: torch.dtype, cp: int | None = None) -> torch.Tensor:
    """logical (B,C,H,W) tensor with any strides -> physical (B,H,W,Cp) in `dtype`, zero padded."""
    B, C, H, W = x.shape
    cp = pad8(C) if cp is None else cp
    out = torch.empty((B, H, W, cp), device=x.device, dtype=dtype)
    sb, sc, sh, sw = x.stride()
    L.check(L.lib().uig_to_nhwc(_p(x), _dt(x), sb, sc, sh, sw, _p(out), B, C, H, W, cp, _dt(out), _stream()), "uig_to_nhwc")
    return out


def from_nhwc(xp: torch.Tensor, C: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """physical (B,H,W,Cp) -> contiguous logical (B,C,H,W) in `dtype`."""
    B, H, W, cp = xp.shape
    out = torch.empty((B, C, H, W), device=xp.device, dtype=dtype)
    sb, sc, sh, sw = out.stride()
    L.check(L.lib().uig_from_nhwc(_p(xp), B, C, H, W, cp, _dt(xp), _p(out), _dt(out), sb, sc, sh, sw, _stream()), "uig_from_nhwc")
    return out


class ToPhysFn(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.C, ctx.in_dtype = x.shape[1], x.dtype
        return to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return from_nhwc(g.contiguous(), ctx.C, ctx.in_dtype), None


class FromPhysFn(Function):
    @staticmethod
    def forward(ctx, xp, C, dtype):
        ctx.cp, ctx.pdtype = xp.shape[3], xp.dtype
        return from_nhwc(xp, C, dtype)

    @staticmethod
    def backward(ctx, g):
        return to_nhwc(g, ctx.pdtype, ctx.cp), None, None


# ----------------------------------------------------------------------------------------- convolution
class ConvSpec:
    """Static description of one Conv2d / ConvTranspose2d layer (+ fused reflection pad and epilogue activation)."""

    def __init__(self, kind, cin, cout, k, stride=1, pad=0, pad_mode="zero", act=L.ACT_NONE, slope=0.0):
        assert kind in ("conv", "convT") and pad_mode in ("zero", "reflect")
        self.kind, self.cin, self.cout, self.k, self.stride, self.pad = kind, cin, cout, k, stride, pad
        self.reflect = pad_mode == "reflect"
        self.act, self.slope = act, slope
        self.cin_p = pad8(cin)
        self.cout_p = pad8(cout)                       # channel count of dy as the backward kernels need it
        self.cout_store = 1 if cout == 1 else self.cout_p   # physical channels of the forward output
        if kind == "convT":
            assert stride == 2 and pad == 1 and k == 3 and not self.reflect, "convT: k3 s2 p1 output_padding1 only"

    def out_hw(self, H, W):
        if self.kind == "conv":
            return (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1
        return 2 * H, 2 * W

    def weight_shape(self):
        return (self.cout, self.cin, self.k, self.k) if self.kind == "conv" else (self.cin, self.cout, self.k, self.k)


def pack_weights(spec: ConvSpec, weight: torch.Tensor, dtype: torch.dtype, wp_fwd: torch.Tensor, wp_dgrad: torch.Tensor):
    """fp32 torch-layout weight -> the two kernel-side operands [rows][tap][cols] (forward and input-gradient)."""
    lib, s, k = L.lib(), _stream(), spec.k
    D0, D1 = weight.shape[0], weight.shape[1]
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    if spec.kind == "conv":      # fwd rows = Cout (dim0), cols = Cin;  dgrad rows = Cin (dim1), cols = Cout
        L.check(lib.uig_pack_weight(_p(weight), _p(wp_fwd), D0, D1, k, k, L.PACK_ROW_DIM0, 0, spec.cout, spec.cin_p, dt, s), "uig_pack_weight")
        L.check(lib.uig_pack_weight(_p(weight), _p(wp_dgrad), D0, D1, k, k, L.PACK_ROW_DIM1, 0, spec.cin, spec.cout_p, dt, s), "uig_pack_weight")
    else:                        # weight (Cin, Cout, k, k): fwd rows = Cout (dim1); dgrad rows = Cin (dim0)
        L.check(lib.uig_pack_weight(_p(weight), _p(wp_fwd), D0, D1, k, k, L.PACK_ROW_DIM1, 0, spec.cout, spec.cin_p, dt, s), "uig_pack_weight")
        L.check(lib.uig_pack_weight(_p(weight), _p(wp_dgrad), D0, D1, k, k, L.PACK_ROW_DIM0, 0, spec.cin, spec.cout_p, dt, s), "uig_pack_weight")


class MultiPacker:
    """Packs every ConvLayer of a set of networks in ONE kernel launch (uig_pack_weights_multi2: both operands of a layer from one read).  The descriptor table holds
    absolute device pointers, so it must be rebuilt if parameters or packed buffers are re-allocated (`valid()` checks)."""

    def __init__(self, layers):
        import numpy as np
        self.layers = list(layers)
        # one record per LAYER: both kernel-side operands come out of one pass over the fp32 weights (uig_pack_weights_multi2)
        dt = np.dtype([("w", "<u8"), ("dst", "<u8"), ("dst2", "<u8"), ("D0", "<i4"), ("D1", "<i4"), ("taps", "<i4"), ("row_dim", "<i4"),
                       ("cols_p", "<i4"), ("cols2_p", "<i4"), ("work_end", "<i8")])
        assert dt.itemsize == 56
        recs, end = [], 0
        for l in self.layers:
            sp, w = l.spec, l.weight
            t = sp.k * sp.k
            D0, D1 = w.shape[0], w.shape[1]
            # conv: fwd rows = dim0 (Cout), dgrad rows = dim1 (Cin); convT: the other way round
            rd = L.PACK_ROW_DIM0 if sp.kind == "conv" else L.PACK_ROW_DIM1
            assert l.wp_fwd.numel() == sp.cout * t * sp.cin_p and l.wp_dgrad.numel() == sp.cin * t * sp.cout_p
            nt = int(L.lib().uig_pack_tiles2(D0, D1, sp.k, sp.k, rd, sp.cin_p, sp.cout_p))
            if nt <= 0:
                raise ValueError(f"uig_pack_tiles2: unsupported weight shape {tuple(w.shape)}")
            end += nt
            recs.append((w.data_ptr(), l.wp_fwd.data_ptr(), l.wp_dgrad.data_ptr(), D0, D1, t, rd, sp.cin_p, sp.cout_p, end))
        self.total = end
        self.ptrs = [(l.weight.data_ptr(), l.wp_fwd.data_ptr(), l.wp_dgrad.data_ptr()) for l in self.layers]
        arr = np.array(recs, dtype=dt)
        self.dtype = self.layers[0].compute_dtype
        self.items = torch.from_numpy(arr.view(np.uint8).copy()).to(self.layers[0].weight.device)
        self.n = len(recs)
        # fp8 layers: the MX quantisation of both packed operands of every such layer as ONE more launch (uig_mx_quantize_multi)
        qdt = np.dtype([("x", "<u8"), ("q", "<u8"), ("s", "<u8"), ("n8", "<i8"), ("block_end", "<i8")])
        qrecs, qend = [], 0
        self.qptrs = []
        for l in self.layers:
            if l.fp8:
                for wp, wq, ws in ((l.wp_fwd, l.wq_fwd, l.ws_fwd), (l.wp_dgrad, l.wq_dgrad, l.ws_dgrad)):
                    n8 = wp.numel() // 8
                    qend += (n8 + 255) // 256
                    qrecs.append((wp.data_ptr(), wq.data_ptr(), ws.data_ptr(), n8, qend))
                    self.qptrs.append((wq, ws))
        self.qn, self.qblocks = len(qrecs), qend
        self.qitems = torch.from_numpy(np.array(qrecs, dtype=qdt).view(np.uint8).copy()).to(self.layers[0].weight.device) if qrecs else None

    def valid(self):
        return all(p == (l.weight.data_ptr(), l.wp_fwd.data_ptr(), l.wp_dgrad.data_ptr()) for p, l in zip(self.ptrs, self.layers)) and \
            self.qn == 2 * sum(1 for l in self.layers if l.fp8) and \
            all(a is b and c is d for (a, c), (b, d) in zip(self.qptrs, [(w, s_) for l in self.layers if l.fp8
                                                                         for w, s_ in ((l.wq_fwd, l.ws_fwd), (l.wq_dgrad, l.ws_dgrad))]))

    def run(self):
        dt = L.BF16 if self.dtype == torch.bfloat16 else L.F32
        L.check(L.lib().uig_pack_weights_multi2(_p(self.items), self.n, self.total, dt, _stream()), "uig_pack_weights_multi2")
        if self.qitems is not None:
            L.check(L.lib().uig_mx_quantize_multi(_p(self.qitems), self.qn, self.qblocks, _stream()), "uig_mx_quantize_multi")
        for l in self.layers:
            l._packed_version = l.weight._version


def packed_shapes(spec: ConvSpec):
    t = spec.k * spec.k
    return (spec.cout, t, spec.cin_p), (spec.cin, t, spec.cout_p)


def _gather(x, wp, bias, y, B, H, W, C, nrows, spec, stride, pad, pm, mode, Ho, Wo, ldc, act, slope, what, pair=None,
            in_partial=None, border_add=None, res_add=None, bst=None, fin=None):
    """one uig_conv_gather launch; pair = (wp2, bias2, group_images) makes it a two-network launch; in_partial receives the
    fused InstanceNorm statistics partials; res_add (a tensor of y's shape) is added to the output in the epilogue;
    fin = (stats, eps, tickets pointer or None): the statistics come out final (uig_conv_gather_fin)"""
    lib = L.lib()
    if fin is not None and bst is None:
        wp2, bias2, g = pair if pair is not None else (None, None, 0)
        rc = lib.uig_conv_gather_fin(_p(x), _p(wp), _p(bias), _p(wp2), _p(bias2), g, _p(in_partial), _p(border_add), _p(res_add), _p(y), B, H, W, C, nrows,
                                     spec.k, spec.k, stride, pad, pm, mode, Ho, Wo, ldc, ldc, act, slope, _dt(x), _p(fin[0]), fin[1], fin[2], _stream())
    elif bst is not None:      # (x_in, stats, act, slope, partial) of the InstanceNorm backward that consumes y as its dy
        wp2, bias2, g = pair if pair is not None else (None, None, 0)
        rc = lib.uig_conv_gather_bst(_p(x), _p(wp), _p(bias), _p(wp2), _p(bias2), g, _p(in_partial), _p(border_add), _p(res_add), _p(y), B, H, W, C, nrows,
                                     spec.k, spec.k, stride, pad, pm, mode, Ho, Wo, ldc, ldc, act, slope, _dt(x),
                                     _p(bst[0]), _p(bst[1]), bst[2], bst[3], _p(bst[4]), _stream())
    elif in_partial is not None or border_add is not None or res_add is not None:
        wp2, bias2, g = pair if pair is not None else (None, None, 0)
        rc = lib.uig_conv_gather_ex(_p(x), _p(wp), _p(bias), _p(wp2), _p(bias2), g, _p(in_partial), _p(border_add), _p(res_add), _p(y), B, H, W, C, nrows,
                                    spec.k, spec.k, stride, pad, pm, mode, Ho, Wo, ldc, ldc, act, slope, _dt(x), _stream())
    elif pair is None:
        rc = lib.uig_conv_gather(_p(x), _p(wp), _p(bias), _p(y), B, H, W, C, nrows, spec.k, spec.k, stride, pad, pm, mode,
                                 Ho, Wo, ldc, ldc, act, slope, _dt(x), _stream())
    else:
        wp2, bias2, g = pair
        rc = lib.uig_conv_gather_pair(_p(x), _p(wp), _p(bias), _p(wp2), _p(bias2), g, _p(y), B, H, W, C, nrows, spec.k, spec.k,
                                      stride, pad, pm, mode, Ho, Wo, ldc, ldc, act, slope, _dt(x), _stream())
    L.check(rc, what)


def mx_quantize(t: torch.Tensor):
    """MX block-scaled fp8 (BASELINE configs[4]): contiguous (..., C) bf16 / f32 tensor, C % 32 == 0 -> (q uint8 (..., C) e4m3
    bytes, s uint8 (..., C/32) E8M0 scale bytes), quantised along the last axis in blocks of 32 (uig_mx_quantize)."""
    if not (t.is_cuda and t.is_contiguous() and t.shape[-1] % 32 == 0):
        raise ValueError(f"mx_quantize: expected a contiguous CUDA tensor with C % 32 == 0, got {tuple(t.shape)}")
    C = t.shape[-1]
    q = torch.empty(t.shape, device=t.device, dtype=torch.uint8)
    s = torch.empty(t.shape[:-1] + (C // 32,), device=t.device, dtype=torch.uint8)
    L.check(L.lib().uig_mx_quantize(_p(t), _p(q), _p(s), t.numel() // C, C, _dt(t), _stream()), "uig_mx_quantize")
    return q, s


def _mx_operand(t: torch.Tensor):
    """(q, s) of an activation / gradient tensor: the pair its producing InstanceNorm launch attached (`_uig_mx`, fused
    quantisation), else a stand-alone quantiser pass"""
    pre = getattr(t, "_uig_mx", None)
    if pre is not None and pre[0].shape == t.shape:
        return pre
    return mx_quantize(t)


def mx_applicable(spec: ConvSpec, B: int, H: int, W: int) -> bool:
    """can this layer's forward / input gradient run on the MX fp8 kernel? (3x3 stride-1 pad-1 conv, 128-multiples of channels)"""
    return (spec.kind == "conv" and spec.k == 3 and spec.stride == 1 and spec.pad == 1 and spec.act == L.ACT_NONE
            and spec.cin_p == spec.cin and spec.cout_p == spec.cout
            and L.lib().uig_conv3x3_mx_fp8_applicable(B, H, W, spec.cin, spec.cout) == 1)


def _conv3x3_mx(xq, xs, mx, bias, pair_bias, group, y, nrows, pad_mode, gather_mode, act, slope, in_partial=None, border_add=None, res_add=None, bst=None):
    """one uig_conv3x3_mx_fp8 launch; mx = (wq, ws) or (wq, ws, wq2, ws2) for a paired launch"""
    B, H, W, C = xq.shape
    wq2, ws2 = (mx[2], mx[3]) if len(mx) == 4 else (None, None)
    L.check(L.lib().uig_conv3x3_mx_fp8(_p(xq), _p(xs), _p(mx[0]), _p(mx[1]), _p(bias), _p(wq2), _p(ws2), _p(pair_bias), group,
                                       _p(in_partial), _p(border_add), _p(res_add), _p(y), B, H, W, C, nrows, pad_mode, gather_mode,
                                       y.shape[3], act, slope,
                                       _p(bst[0]) if bst else None, _p(bst[1]) if bst else None, bst[2] if bst else 0, bst[3] if bst else 0.0,
                                       _p(bst[4]) if bst else None, _stream()), "uig_conv3x3_mx_fp8")


def in_stats_fusable(spec: ConvSpec, H: int, W: int, B: int = 1, dtype: torch.dtype = torch.bfloat16) -> bool:
    """can this layer's forward launch also emit the statistics of the InstanceNorm that follows it? (rule of uig_conv_gather_ex)"""
    Ho, Wo = spec.out_hw(H, W)
    grid = Ho * Wo if spec.kind == "conv" else (Ho // spec.stride) * (Wo // spec.stride)
    if spec.cout % 64 != 0 or grid % 64 != 0 or spec.act != L.ACT_NONE:
        return False
    if spec.cout > 64:
        return True
    # 64 output channels: only the phase-fused transposed kernel has the epilogue for it
    return spec.kind == "convT" and dtype == torch.bfloat16 and \
        L.lib().uig_conv_tr2_applicable(B, H, W, spec.cin_p, spec.cout, spec.cout_store, spec.cout_store, L.BF16) == 1


def conv_forward(spec: ConvSpec, x: torch.Tensor, wp_fwd: torch.Tensor, bias: torch.Tensor | None, pair=None,
                 want_in_stats: bool = False, mx=None, in_eps: float = 1e-5) -> torch.Tensor:
    """want_in_stats: also accumulate the following InstanceNorm's (sum, sum^2) partials in the epilogue; they travel to
    ops.InstNormActFn as the attribute `_uig_in_partial` of the returned tensor.
    mx = (wq, ws[, wq2, ws2]): run the layer on the MX block-scaled fp8 kernel (x is quantised here, the output stays bf16)."""
    _chk_phys(x, "conv_forward")
    B, H, W, C = x.shape
    if C != spec.cin_p:
        raise ValueError(f"conv_forward: input has {C} physical channels, layer expects {spec.cin_p}")
    Ho, Wo = spec.out_hw(H, W)
    y = torch.empty((B, Ho, Wo, spec.cout_store), device=x.device, dtype=x.dtype)
    if spec.kind == "conv":
        mode, pm = L.GATHER_DIRECT, (L.PAD_REFLECT if spec.reflect else L.PAD_ZERO)
    else:
        mode, pm = L.GATHER_TRANSPOSED, L.PAD_ZERO
    part = None
    if want_in_stats and in_stats_fusable(spec, H, W, B, x.dtype):
        nslab = Ho * Wo // 64
        part = torch.empty((B * nslab * spec.cout_store * 2,), device=x.device, dtype=torch.float32)
    fin = None
    if mx is not None:
        xq, xs = _mx_operand(x)
        _conv3x3_mx(xq, xs, mx, bias, pair[1] if pair is not None else None, pair[2] if pair is not None else 0, y, spec.cout, pm, mode,
                    spec.act, spec.slope, part)
    else:
        # round 4: (mean, rstd) of the following norm (eps = in_eps) come out of this launch final - except where the inference norm
        # finalises inside its own apply launch (small batches: InstNormAct.forward -> instnorm_infer)
        infer_fused = INFER_FUSED_IN and not torch.is_grad_enabled() and B <= INFER_FUSED_MAX_BATCH and Ho * Wo // 64 <= INFER_FUSED_MAX_PARTIALS
        if part is not None and IN_TICKETS and not infer_fused:
            fin = (torch.empty((B, spec.cout_store, 2), device=x.device, dtype=torch.float32), float(in_eps), _tickets(x.device, 0, B))
        _gather(x, wp_fwd, bias, y, B, H, W, C, spec.cout, spec, spec.stride, spec.pad, pm, mode, Ho, Wo, spec.cout_store,
                spec.act, spec.slope, "uig_conv_gather(fwd)", pair, part, fin=fin)
    if part is not None:
        y._uig_in_partial = (part, Ho * Wo // 64)
    if fin is not None:
        y._uig_in_stats = (fin[0], fin[1])
    return y


def _dy_padded(spec: ConvSpec, dy: torch.Tensor) -> torch.Tensor:
    """dy as the backward kernels want it: (B,Ho,Wo,cout_p) contiguous (pads the 1-channel discriminator head)."""
    if dy.shape[3] == spec.cout_p and dy.is_contiguous():
        return dy
    B, Ho, Wo, c = dy.shape
    return to_nhwc(dy.permute(0, 3, 1, 2), dy.dtype, spec.cout_p)


def conv_dgrad(spec: ConvSpec, dy: torch.Tensor, wp_dgrad: torch.Tensor, in_hw, pair=None, res_add=None, mx=None, bst=None) -> torch.Tensor:
    """aten::convolution_backward, input gradient.  dy: (B,Ho,Wo,cout_p).  pair = (wp_dgrad2, None, group_images).
    res_add: a second gradient of the input (the ResBlock skip path's) to be summed in: fused into the launch's epilogue
    where the kernel supports it, one in-place add otherwise.
    bst = (x_in, stats, act, slope) of the InstanceNorm whose backward will consume the returned dx as its dy: where the launch
    supports it (bf16 strip / fp8 kernel with the border terms), its backward statistics come out of this launch's epilogue and
    travel on dx as `_uig_bst_partial`."""
    B, Ho, Wo, Cd = dy.shape
    H, W = in_hw
    s = _stream()
    # pad-1 reflection 3x3 on 64-wide bf16 maps (the ResBlock convs at 256x256): the persistent strip kernel folds the mirrored
    # terms itself (mirror pixels, uig_reflect3x3_dgrad_mirror) - one launch, square map or not
    mirror = (spec.kind == "conv" and spec.reflect and spec.k == 3 and spec.pad == 1 and spec.stride == 1 and REFLECT_DGRAD_DIRECT
              and mx is None and spec.cin_p == spec.cin and not (bst is not None and FUSE_BWD_STATS and not MIRROR_BST)
              and L.lib().uig_reflect3x3_dgrad_mirror_applicable(B, Ho, Wo, Cd, spec.cin, spec.cin_p, _dt(dy)) == 1)
    # the same on the MX fp8 kernel (round 3): mirror pixels re-quantised in LDS, no bf16 border GEMM in front
    mx_mirror = (mx is not None and MX_DGRAD_MIRROR and spec.kind == "conv" and spec.reflect and spec.k == 3 and spec.pad == 1 and spec.stride == 1
                 and REFLECT_DGRAD_DIRECT and spec.cin_p == spec.cin and not (bst is not None and FUSE_BWD_STATS) and dy.dtype == torch.bfloat16
                 and L.lib().uig_conv3x3_mx_fp8_dgrad_mirror_applicable(B, Ho, Wo, Cd, spec.cin) == 1)
    if res_add is not None:
        fusable = (spec.kind == "conv" and spec.reflect and spec.k == 3 and spec.pad == 1 and spec.stride == 1 and ((H == W and 4 <= H <= 128) or mirror or mx_mirror)
                   and REFLECT_DGRAD_DIRECT and FUSE_SKIP_GRAD and res_add.is_contiguous() and res_add.dtype == dy.dtype
                   and tuple(res_add.shape) == (B, H, W, spec.cin_p)
                   and L.lib().uig_conv_strip_applicable(B, Ho, Wo, Cd, spec.cin, H, W, -1, 1, _dt(dy)) == 1)
        if not fusable:
            dx = conv_dgrad(spec, dy, wp_dgrad, in_hw, pair, mx=mx)
            return dx.add_(res_add)
    if spec.kind == "convT":     # gradient of a transposed conv = strided direct conv of dy
        dx = torch.empty((B, H, W, spec.cin_p), device=dy.device, dtype=dy.dtype)
        _gather(dy, wp_dgrad, None, dx, B, Ho, Wo, Cd, spec.cin, spec, spec.stride, spec.pad, L.PAD_ZERO, L.GATHER_DIRECT, H, W,
                spec.cin_p, L.ACT_NONE, 0.0, "uig_conv_gather(convT dgrad)", pair)
        return dx
    if mx_mirror:
        wq2, ws2 = (mx[2], mx[3]) if len(mx) == 4 else (None, None)
        dx = torch.empty((B, H, W, spec.cin_p), device=dy.device, dtype=dy.dtype)
        dq, ds = _mx_operand(dy)
        L.check(L.lib().uig_conv3x3_mx_fp8_dgrad_mirror(_p(dq), _p(ds), _p(mx[0]), _p(mx[1]), _p(wq2), _p(ws2), pair[2] if pair is not None else 0,
                                                        _p(res_add), _p(dx), B, Ho, Wo, Cd, spec.cin, spec.cin_p, s), "uig_conv3x3_mx_fp8_dgrad_mirror")
        return dx
    if mirror or (spec.reflect and spec.k == 3 and spec.pad == 1 and spec.stride == 1 and H == W and 4 <= H <= 128 and REFLECT_DGRAD_DIRECT
                  and L.lib().uig_conv_strip_applicable(B, Ho, Wo, Cd, spec.cin, H, W, -1, 1, _dt(dy)) == 1):
        # pad-1 reflection, 3x3: dx on the exact H x W grid (zero-pad transposed conv: whole tile rounds on 256 CUs) plus the
        # mirrored-border terms from one small 8-phase GEMM, added in the strip kernel's epilogue.  No (H+2)x(W+2)
        # padded gradient, no fold kernel.
        lib = L.lib()
        wp2, g = (pair[0], pair[2]) if pair is not None else (None, 0)
        dx = torch.empty((B, H, W, spec.cin_p), device=dy.device, dtype=dy.dtype)
        bord = None
        if not mirror:
            bord = torch.empty((B, 8, H, spec.cin_p), device=dy.device, dtype=dy.dtype)
            L.check(lib.uig_reflect3x3_dgrad_border(_p(dy), _p(wp_dgrad), _p(wp2), g, _p(bord), B, Ho, Wo, Cd, spec.cin, spec.cin_p,
                                                    _dt(dy), s), "uig_reflect3x3_dgrad_border")
        bpart = gm = None
        if bst is not None and (MIRROR_BST if mirror else FUSE_BWD_STATS) and dy.dtype == torch.bfloat16 and (H * W) % 64 == 0 and spec.cin % 64 == 0 \
                and spec.cin_p == spec.cin and tuple(bst[0].shape) == (B, H, W, spec.cin_p) and bst[0].dtype == dy.dtype and bst[0].is_contiguous():
            bpart = torch.empty((B * (H * W // 64) * spec.cin_p * 2,), device=dy.device, dtype=torch.float32)
            bst = (bst[0], bst[1], bst[2], bst[3], bpart)
        else:
            bst = None
        if mx is not None:       # main term on the MX fp8 kernel (dy quantised here); the mirrored-border GEMM above stays bf16
            dq, ds = _mx_operand(dy)
            _conv3x3_mx(dq, ds, mx, None, None, pair[2] if pair is not None else 0, dx, spec.cin, L.PAD_ZERO, L.GATHER_TRANSPOSED,
                        L.ACT_NONE, 0.0, None, bord, res_add, bst)
        elif mirror and bst is not None:
            # round 4: the launch also emits - final - the statistics of the InstanceNorm backward that consumes dx
            gm = torch.empty((B, spec.cin_p, 2), device=dy.device, dtype=torch.float32)
            L.check(lib.uig_reflect3x3_dgrad_mirror_bst(_p(dy), _p(wp_dgrad), _p(wp2), g, _p(res_add), _p(dx), B, Ho, Wo, Cd, spec.cin, spec.cin_p, _dt(dy),
                                                        _p(bst[0]), _p(bst[1]), bst[2], bst[3], _p(bpart), _p(gm), _tickets(dy.device, 1, B), s),
                    "uig_reflect3x3_dgrad_mirror_bst")
        elif mirror:
            L.check(lib.uig_reflect3x3_dgrad_mirror(_p(dy), _p(wp_dgrad), _p(wp2), g, _p(res_add), _p(dx), B, Ho, Wo, Cd, spec.cin, spec.cin_p,
                                                    _dt(dy), s), "uig_reflect3x3_dgrad_mirror")
        else:
            _gather(dy, wp_dgrad, None, dx, B, Ho, Wo, Cd, spec.cin, spec, 1, 1, L.PAD_ZERO, L.GATHER_TRANSPOSED, H, W, spec.cin_p,
                    L.ACT_NONE, 0.0, "uig_conv_gather(dgrad+border)", pair, None, bord, res_add, bst)
        if gm is not None:
            dx._uig_bst_gm = gm
        elif bpart is not None:
            dx._uig_bst_partial = (bpart, H * W // 64)
        return dx
    if spec.reflect:             # gradient w.r.t. the reflection-padded input, then fold the border back
        P = spec.pad
        dxp = torch.empty((B, H + 2 * P, W + 2 * P, spec.cin_p), device=dy.device, dtype=dy.dtype)
        _gather(dy, wp_dgrad, None, dxp, B, Ho, Wo, Cd, spec.cin, spec, spec.stride, 0, L.PAD_ZERO, L.GATHER_TRANSPOSED,
                H + 2 * P, W + 2 * P, spec.cin_p, L.ACT_NONE, 0.0, "uig_conv_gather(dgrad)", pair)
        dx = torch.empty((B, H, W, spec.cin_p), device=dy.device, dtype=dy.dtype)
        L.check(L.lib().uig_reflect_fold(_p(dxp), _p(dx), B, H, W, spec.cin_p, P, _dt(dy), s), "uig_reflect_fold")
        return dx
    dx = torch.empty((B, H, W, spec.cin_p), device=dy.device, dtype=dy.dtype)
    _gather(dy, wp_dgrad, None, dx, B, Ho, Wo, Cd, spec.cin, spec, spec.stride, spec.pad, L.PAD_ZERO, L.GATHER_TRANSPOSED, H, W,
            spec.cin_p, L.ACT_NONE, 0.0, "uig_conv_gather(dgrad)", pair)
    return dx


_WGRAD_BLOCKS = int(os.environ.get("UIG_WGRAD_BLOCKS", "512"))   # target grid of the split-K weight-gradient kernel (2 blocks per CU)


def _wgrad_operands(spec: ConvSpec, x: torch.Tensor, dy: torch.Tensor):
    """(dense P, gathered Q, Mh, Mw, Np, Hq, Wq, Cq, pad mode, D0, D1) of the weight-gradient GEMM"""
    B, H, W, _ = x.shape
    _, Ho, Wo, _ = dy.shape
    if spec.kind == "conv":      # dense = dy, gathered = x;  dW (Cout, Cin, k, k)
        return dy, x, Ho, Wo, spec.cout_p, H, W, spec.cin_p, (L.PAD_REFLECT if spec.reflect else L.PAD_ZERO), spec.cout, spec.cin
    return x, dy, H, W, spec.cin_p, Ho, Wo, spec.cout_p, L.PAD_ZERO, spec.cin, spec.cout      # convT: dense = x, gathered = dy


def conv_wgrad_pair_partial(spec: ConvSpec, x: torch.Tensor, dy: torch.Tensor, group: int):
    """Partial weight-gradient slabs of TWO networks (first `group` images / the rest) in one launch.
    Returns [(workspace of network i, splits)] for conv_wgrad(partial=...)."""
    lib = L.lib()
    B, k = x.shape[0], spec.k
    Pt, Qt, Mh, Mw, Np, Hq, Wq, Cq, pm, _, _ = _wgrad_operands(spec, x, dy)
    splits = int(lib.uig_wgrad_pair_splits(B, group, Mh, Mw, Np, Hq, Wq, Cq, k, k, spec.stride, spec.pad, _dt(x), _WGRAD_BLOCKS))
    if splits <= 0:
        return None
    per = splits * Np * k * k * Cq
    ws = torch.empty((2 * per,), device=x.device, dtype=torch.float32)
    L.check(lib.uig_wgrad_partial_pair(_p(Pt), _p(Qt), _p(ws), B, group, Mh, Mw, Np, Hq, Wq, Cq, k, k, spec.stride, spec.pad, pm,
                                       splits, _dt(x), _stream()), "uig_wgrad_partial_pair")
    return [(ws[:per], splits), (ws[per:], splits)]


def conv_wgrad(spec: ConvSpec, x: torch.Tensor, dy: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False,
               bias_rider=None, partial=None) -> torch.Tensor:
    """aten::convolution_backward, weight gradient (fp32, torch layout).  With `out` the split-K reduce writes (or, with
    accumulate=True, adds) straight into that tensor, e.g. the layer's slice of the flat gradient buffer.
    partial = (workspace, splits) from conv_wgrad_pair_partial: only the reduce runs."""
    lib, s = L.lib(), _stream()
    B, k = x.shape[0], spec.k
    Pt, Qt, Mh, Mw, Np, Hq, Wq, Cq, pm, D0, D1 = _wgrad_operands(spec, x, dy)
    if partial is not None:
        ws, splits = partial
    else:
        splits = int(lib.uig_wgrad_splits(B, Mh, Mw, Np, Hq, Wq, Cq, k, k, spec.stride, spec.pad, _dt(x), _WGRAD_BLOCKS))
        ws = torch.empty((splits * Np * k * k * Cq,), device=x.device, dtype=torch.float32)
        L.check(lib.uig_wgrad_partial(_p(Pt), _p(Qt), _p(ws), B, Mh, Mw, Np, Hq, Wq, Cq, k, k, spec.stride, spec.pad, pm,
                                      splits, _dt(x), s), "uig_wgrad_partial")
    if out is None:
        out, accumulate = torch.empty(spec.weight_shape(), device=x.device, dtype=torch.float32), False
    elif not (out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == tuple(spec.weight_shape())):
        raise ValueError("conv_wgrad: `out` must be a contiguous fp32 tensor of the weight's shape")
    if bias_rider is not None:      # (colsum tuple, first image, images, db, accumulate_db): bias gradient rides on the reduce launch
        (cpart, slabs_per_img, C), img0, nimg, db, acc_b = bias_rider
        sub = cpart[img0 * slabs_per_img * C * 2:]
        L.check(lib.uig_wgrad_reduce_bias(_p(ws), _p(out), Np, Cq, k * k, splits, D0, D1, 1 if accumulate else 0, _p(sub),
                                          nimg * slabs_per_img, C, spec.cout, _p(db), 1 if acc_b else 0, s), "uig_wgrad_reduce_bias")
    else:
        L.check(lib.uig_wgrad_reduce(_p(ws), _p(out), Np, Cq, k * k, splits, D0, D1, 1 if accumulate else 0, s), "uig_wgrad_reduce")
    return out


def bias_grad(dy: torch.Tensor, nreal: int, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    B, Ho, Wo, C = dy.shape
    ws = torch.empty((int(L.lib().uig_colsum_workspace_floats(C)),), device=dy.device, dtype=torch.float32)
    if out is None:
        out, accumulate = torch.empty((nreal,), device=dy.device, dtype=torch.float32), False
    L.check(L.lib().uig_bias_grad(_p(dy), _p(out), _p(ws), B * Ho * Wo, C, nreal, 1 if accumulate else 0, _dt(dy), _stream()), "uig_bias_grad")
    return out


def _bias_grad_from_partials(cs, img0, nimg, nreal, out, accumulate):
    cpart, slabs_per_img, C = cs
    sub = cpart[img0 * slabs_per_img * C * 2:]
    if out is None:
        out, accumulate = torch.empty((nreal,), device=cpart.device, dtype=torch.float32), False
    L.check(L.lib().uig_bias_grad_from_partials(_p(sub), _p(out), nimg * slabs_per_img, C, nreal, 1 if accumulate else 0, _stream()),
            "uig_bias_grad_from_partials")
    return out


def _param_grads(layer, spec, x, dy, need_w, need_b, colsum=None, img0=0, partial=None):
    """dW / db of one layer.  When the parameter already owns a .grad buffer (the trainer's flat gradient buffer, or any
    earlier backward) the reduce kernels ADD into it in place and autograd gets None (= nothing more to accumulate):
    gradient-accumulation fusion, no temporary dW and no extra add kernel.  Otherwise they are returned the usual way."""
    dW = db = None
    b = layer.bias
    rider = None
    if need_w and need_b and colsum is not None and colsum[2] == dy.shape[3] and layer.fuse_grad_accum and b.grad is not None \
            and b.grad.is_contiguous():
        rider, need_b = (colsum, img0, dy.shape[0], b.grad, True), False       # bias gradient finished by the reduce launch
    if need_w:
        w = layer.weight
        if layer.fuse_grad_accum and w.grad is not None and w.grad.is_contiguous():
            conv_wgrad(spec, x, dy, out=w.grad, accumulate=True, bias_rider=rider, partial=partial)
        else:
            dW = conv_wgrad(spec, x, dy, bias_rider=rider, partial=partial)
    if need_b:
        b = layer.bias
        fused = layer.fuse_grad_accum and b.grad is not None and b.grad.is_contiguous()
        if colsum is not None and colsum[2] == dy.shape[3]:      # column sums already produced by the InstanceNorm backward
            r = _bias_grad_from_partials(colsum, img0, dy.shape[0], spec.cout, b.grad if fused else None, fused)
            db = None if fused else r
        elif fused:
            bias_grad(dy, spec.cout, out=b.grad, accumulate=True)
        else:
            db = bias_grad(dy, spec.cout)
    return dW, db


def _param_grads_pair(layers, spec, x, dy, group, colsum, pparts, bias=True, colsum2=None):
    """Both networks' dW (+ db) from a paired partial workspace in ONE reduce launch.  Only the training configuration
    (gradients accumulated in place into existing contiguous .grad buffers of both layers); returns False otherwise."""
    l1, l2 = layers
    for l in layers:
        if not (l.fuse_grad_accum and l.weight.grad is not None and l.weight.grad.is_contiguous()
                and l.bias.grad is not None and l.bias.grad.is_contiguous()):
            return False
    (ws1, splits), (ws2, _) = pparts
    if ws2.data_ptr() != ws1.data_ptr() + ws1.numel() * 4:
        return False
    k = spec.k
    _, _, _, _, Np, _, _, Cq, _, D0, D1 = _wgrad_operands(spec, x, dy)
    rider = bias and colsum is not None and colsum[2] == dy.shape[3]
    lib = L.lib()
    if rider and colsum2 is not None:
        # colsum2 = ((cpart, slabs per image, C), first image / images of network a, first image / images of network b) of the OTHER
        # generator pass, whose weight-gradient partials are in the same workspace (_combined_wgrad)
        cpart, spi, C = colsum
        (cp2, spi2, _), (a0, an), (b0, bn) = colsum2
        L.check(lib.uig_wgrad_reduce_pair2(_p(ws1), _p(l1.weight.grad), _p(l2.weight.grad), Np, Cq, k * k, splits, D0, D1, 1,
                                           _p(cpart), _p(cpart[group * spi * C * 2:]), group * spi, (dy.shape[0] - group) * spi,
                                           _p(cp2[a0 * spi2 * C * 2:]), _p(cp2[b0 * spi2 * C * 2:]), an * spi2, bn * spi2,
                                           C, spec.cout, _p(l1.bias.grad), _p(l2.bias.grad), 1, _stream()), "uig_wgrad_reduce_pair2")
    elif rider:
        cpart, spi, C = colsum
        ca, cb = cpart, cpart[group * spi * C * 2:]
        L.check(lib.uig_wgrad_reduce_pair(_p(ws1), _p(l1.weight.grad), _p(l2.weight.grad), Np, Cq, k * k, splits, D0, D1, 1, _p(ca), _p(cb),
                                          group * spi, (dy.shape[0] - group) * spi, C, spec.cout, _p(l1.bias.grad), _p(l2.bias.grad), 1,
                                          _stream()), "uig_wgrad_reduce_pair")
    else:
        L.check(lib.uig_wgrad_reduce_pair(_p(ws1), _p(l1.weight.grad), _p(l2.weight.grad), Np, Cq, k * k, splits, D0, D1, 1, None, None,
                                          0, 0, 0, 0, None, None, 0, _stream()), "uig_wgrad_reduce_pair")
        if bias:
            for l, dys in ((l1, dy[:group]), (l2, dy[group:])):
                bias_grad(dys, spec.cout, out=l.bias.grad, accumulate=True)
    return True


class SkipLink:
    """Side channel of one ResBlock call: the InstanceNorm that adds the residual hands the skip path's gradient (= its incoming
    gradient) to the block's FIRST convolution, whose input-gradient launch sums it in, instead of returning it to autograd
    (which would add the two gradients of the block input with a separate full-tensor kernel).  Ordering is given by the
    graph: the first conv's backward can only run after that norm's backward."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None


def _conv_backward(ctx, dy, layers, group):
    """Shared backward of ConvFn / PairConvFn.  The input gradient and the parameter gradients are independent given dy:
    the parameter-gradient kernels are forked onto a side stream and joined before returning.  The input-gradient grid
    rarely fills a whole number of rounds on 256 CUs (e.g. 288 tiles) and the weight-gradient blocks soak up the idle CUs
    of its tail.  Safe for the caching allocator (and capturable into a HIP graph): the join orders every later use of
    dy / x after the fork."""
    spec = layers[0].spec
    x, y = ctx.saved_tensors
    colsum = getattr(dy, "_uig_colsum", None) if dy.is_contiguous() else None
    dy_mx = getattr(dy, "_uig_mx", None) if dy.is_contiguous() else None
    dy = dy.contiguous()
    if spec.act != L.ACT_NONE:
        colsum = None      # epilogue activation backward on the saved output
        g = torch.empty_like(dy)
        L.check(L.lib().uig_act_bwd(_p(dy), _p(y), _p(g), dy.numel(), spec.act, spec.slope, _dt(dy), _stream()), "uig_act_bwd")
        dy = g
    dy = _dy_padded(spec, dy)
    if dy_mx is not None and spec.act == L.ACT_NONE and dy_mx[0].shape == dy.shape and getattr(dy, "_uig_mx", None) is None:
        dy._uig_mx = dy_mx
    npar = len(layers)
    need_x = ctx.needs_input_grad[0]
    need_w = [ctx.needs_input_grad[1 + 2 * i] for i in range(npar)]
    need_b = [ctx.needs_input_grad[2 + 2 * i] for i in range(npar)]
    any_p = any(need_w) or any(need_b)
    pair = None if npar == 1 else (layers[1].wp_dgrad, None, group)
    mx = None
    if all(l.mx_active(dy.shape[0], ctx.in_hw[0], ctx.in_hw[1]) for l in layers):
        mx = sum(((l.wq_dgrad, l.ws_dgrad) for l in layers), ())
    fused_all = all(l.fuse_grad_accum and l.weight.grad is not None and l.bias.grad is not None for l in layers)
    defer = _DEFER_JOIN.get(torch.device(dy.device).index, False) and any_p and fused_all and PARALLEL_BACKWARD
    par = (need_x or defer) and any_p and PARALLEL_BACKWARD
    main = torch.cuda.current_stream(dy.device)
    link = getattr(ctx, "skip_link", None)
    skip = None
    if link is not None:
        skip, link.grad = link.grad, None
    dx = None
    if need_x and not par:
        dx = conv_dgrad(spec, dy, layers[0].wp_dgrad, ctx.in_hw, pair, skip, mx=mx, bst=getattr(ctx, "bst", None))
    grads = []
    if par:
        side = _side_stream(dy.device)
        side.wait_stream(main)
    with (torch.cuda.stream(side) if par else contextlib.nullcontext()):
        stash = _WG_STASH.get(torch.device(dy.device).index)
        combined = stash is not None and npar == 2 and all(need_w) and all(need_b) and fused_all and PAIR_WGRAD \
            and _combined_wgrad(layers, spec, x, dy, group, colsum, stash)
        pparts = None
        if combined:
            grads, layers_left = [None] * (2 * npar), ()
            if par:      # a stashed batch is read by a later launch on the side stream
                x.record_stream(side); dy.record_stream(side)
                if colsum is not None:
                    colsum[0].record_stream(side)
        else:
            pparts = conv_wgrad_pair_partial(spec, x, dy, group) if (npar == 2 and all(need_w) and PAIR_WGRAD) else None
        if combined:
            pass
        elif pparts and all(need_b) and _param_grads_pair(layers, spec, x, dy, group, colsum, pparts):
            grads, layers_left = [None] * (2 * npar), ()
        else:
            layers_left = layers
        for i, layer in enumerate(layers_left):
            if npar == 1:
                xs, dys, i0 = x, dy, 0
            else:
                xs, dys, i0 = (x[:group], dy[:group], 0) if i == 0 else (x[group:], dy[group:], group)
            grads.extend(_param_grads(layer, spec, xs, dys, need_w[i], need_b[i], colsum, i0, pparts[i] if pparts else None))
    if par:
        if need_x:
            dx = conv_dgrad(spec, dy, layers[0].wp_dgrad, ctx.in_hw, pair, skip, mx=mx, bst=getattr(ctx, "bst", None))
        if defer:
            # no join here: the side stream keeps working behind the main stream's next ops (InstanceNorm backward, the next
            # layer's input gradient, ...).  The tensors it reads are pinned for the allocator with record_stream; the owner
            # of the deferred region joins once (join_param_grads) before anything consumes the .grad buffers.
            x.record_stream(side); dy.record_stream(side)
            if colsum is not None:
                colsum[0].record_stream(side)
        else:
            main.wait_stream(side)
    if skip is not None and dx is None:       # input needs no gradient from the conv itself, the skip path's still flows
        dx = skip
    return (dx, *grads)


class ConvFn(Function):
    """y = act(conv(x, W) + b) on physical NHWC tensors; backward = dgrad / wgrad / bias-grad HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer, skip_link=None):
        spec = layer.spec
        mx = (layer.wq_fwd, layer.ws_fwd) if layer.mx_active(x.shape[0], x.shape[1], x.shape[2]) else None
        y = conv_forward(spec, x, layer.wp_fwd, bias, want_in_stats=layer.emit_in_stats, mx=mx, in_eps=layer.in_eps)
        ctx.layer, ctx.in_hw, ctx.skip_link = layer, (x.shape[1], x.shape[2]), skip_link
        ctx.bst = getattr(x, "_uig_bst", None)      # x is the output of an InstanceNorm: (its input, its stats, act, slope)
        ctx.save_for_backward(x, y if spec.act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        return (*_conv_backward(ctx, dy, (ctx.layer,), 0), None, None)


class PairConvFn(Function):
    """The same layer of TWO networks of identical architecture in one launch: the first `group` images of x go through
    layer1, the rest through layer2 (CycleGAN: G_A / G_B and D_A / D_B always see same-shaped batches).  Halves the launch
    count and doubles the tiles per launch, which is what fills 256 CUs at a per-GPU batch of 4."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, layer1, layer2, group, skip_link=None):
        spec = layer1.spec
        mx = None
        if layer1.mx_active(x.shape[0], x.shape[1], x.shape[2]) and layer2.mx_active(x.shape[0], x.shape[1], x.shape[2]):
            mx = (layer1.wq_fwd, layer1.ws_fwd, layer2.wq_fwd, layer2.ws_fwd)
        y = conv_forward(spec, x, layer1.wp_fwd, b1, pair=(layer2.wp_fwd, b2, group), want_in_stats=layer1.emit_in_stats, mx=mx, in_eps=layer1.in_eps)
        ctx.layers, ctx.group, ctx.in_hw, ctx.skip_link = (layer1, layer2), group, (x.shape[1], x.shape[2]), skip_link
        ctx.bst = getattr(x, "_uig_bst", None)
        ctx.save_for_backward(x, y if spec.act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        return (*_conv_backward(ctx, dy, ctx.layers, ctx.group), None, None, None, None)


# Inference InstanceNorm (statistics finalised inside the apply launch, no finalize launch).  Round 2's form made every block
# re-reduce ALL channels' partials (64 x 256 x 8 bytes): 1.371 vs 0.959 ms at batch 1 - opt-in then.  Round 3: blocks own 64 channels
# and finalise only those (in_apply_fwd_fin_cs_kernel): G9 bf16 1 x 256 x 256 0.800 -> 0.746 ms, bit-identical; a tie at 512 x 512
# (its ResBlock maps have 256 partials: not fused) and 1.5 % slower at batch 8 - so ON by default for batches <= 2 only.
INFER_FUSED_IN = os.environ.get("UIG_INFER_FUSED_IN", "1") != "0"
INFER_FUSED_MAX_BATCH = int(os.environ.get("UIG_INFER_FUSED_MAX_BATCH", "2"))


class small_grid_kernels:
    """Context manager (inference only): let plain 3x3 launches of very small grids (<= 64 blocks of 128x128: batch 1) run on the
    64x64-tile strip kernel while it is active.  Kernel selection is process-global library state, so the region is serialised by a
    lock (a second thread's Translator call waits), the previous mode is restored on exit (nesting keeps the outer choice) and a
    train step must not run concurrently with it - the training path never selects these kernels (their statistics sum in another
    order: the data-parallel step must equal the full-batch step).  Captured HIP graphs keep the choice made at capture time."""

    MODE = int(os.environ.get("UIG_INFER_SMALL_GRID", "0"))      # 0 = auto (default), 1 = wherever it applies, 2 = never (A/B)
    _lock = threading.RLock()
    _current = 2                                                   # the library's default: never (training keeps a batch-independent choice)

    def __enter__(self):
        cls = small_grid_kernels
        cls._lock.acquire()
        self._prev = cls._current
        cls._current = self.MODE
        L.lib().uig_debug_set_strip_small(self.MODE)
        return self

    def __exit__(self, *exc):
        cls = small_grid_kernels
        cls._current = self._prev
        L.lib().uig_debug_set_strip_small(self._prev)
        cls._lock.release()
        return False


def instnorm_infer(x, residual, act, slope, eps):
    """InstanceNorm(+act, +residual) forward with no autograd state (inference): the statistics are
    finalised inside the apply kernel - one launch behind a convolution that emitted the partials, two otherwise."""
    _chk_phys(x, "instnorm")
    B, H, W, C = x.shape
    lib = L.lib()
    y = torch.empty_like(x)
    pre = getattr(x, "_uig_in_partial", None)
    L.check(lib.uig_instnorm_act_fwd_infer(_p(x), _p(residual), _p(y), _p(pre[0]), pre[1], None, B, H * W, C, eps, act, slope,
                                           _dt(x), _stream()), "uig_instnorm_act_fwd_infer")
    return y


INFER_FUSED_MAX_PARTIALS = int(os.environ.get("UIG_INFER_FUSED_MAX_PARTIALS", "64"))


def instnorm_infer_applicable(x) -> bool:
    """every block of the fused kernel re-reads the image's partial statistics (np * C * 8 bytes): only worth it for the small
    maps of the ResBlocks (64 partials per image at 256x256), not for the 128^2 / 256^2 layers (256 / 1024 partials)"""
    pre = getattr(x, "_uig_in_partial", None)
    return pre is not None and x.shape[0] <= INFER_FUSED_MAX_BATCH and pre[1] <= INFER_FUSED_MAX_PARTIALS and \
        pre[0].numel() == x.shape[0] * pre[1] * x.shape[3] * 2


class InstNormActFn(Function):
    @staticmethod
    def forward(ctx, x, residual, act, slope, eps, skip_link=None, mx_fwd=False, mx_bwd=False):
        """mx_fwd / mx_bwd: also emit the MX fp8 form of the output / of the backward's dx (attribute `_uig_mx` of that tensor) for
        the fp8 convolution that consumes it (BASELINE configs[4])"""
        _chk_phys(x, "instnorm")
        B, H, W, C = x.shape
        lib = L.lib()
        ws = torch.empty((int(lib.uig_instnorm_workspace_floats(B, H * W, C)),), device=x.device, dtype=torch.float32)
        stats = torch.empty((B, C, 2), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        pre = getattr(x, "_uig_in_partial", None)
        fin = getattr(x, "_uig_in_stats", None)      # (stats, eps): already final, out of the convolution launch (round 4)
        if fin is not None and not (tuple(fin[0].shape) == (B, C, 2) and fin[1] == float(eps)):
            fin = None
        ctx.mx_bwd = bool(mx_bwd) and x.dtype == torch.bfloat16 and C % 32 == 0
        want_mx = bool(mx_fwd) and x.dtype == torch.bfloat16 and C % 32 == 0
        have_pre = pre is not None and pre[0].numel() == B * pre[1] * C * 2
        if fin is not None or (IN_TICKETS and not have_pre and B <= _TICKET_WORDS):
            q = s = None
            if want_mx:
                q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
                s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
            if fin is not None:
                stats = fin[0]
                L.check(lib.uig_instnorm_apply_fwd(_p(x), _p(residual), _p(y), _p(stats), _p(q), _p(s), B, H * W, C, act, slope, _dt(x), _stream()),
                        "uig_instnorm_apply_fwd")
            else:      # this norm's own statistics pass, finalising itself: two launches
                L.check(lib.uig_instnorm_act_fwd_t(_p(x), _p(residual), _p(y), _p(stats), _p(ws), _tickets(x.device, 0, B), _p(q), _p(s),
                                                   B, H * W, C, eps, act, slope, _dt(x), _stream()), "uig_instnorm_act_fwd_t")
            if want_mx:
                y._uig_mx = (q, s)
        elif mx_fwd and x.dtype == torch.bfloat16 and C % 32 == 0:
            q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
            s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
            have = pre is not None and pre[0].numel() == B * pre[1] * C * 2
            L.check(lib.uig_instnorm_act_fwd_mx(_p(x), _p(residual), _p(y), _p(stats), _p(pre[0]) if have else None, pre[1] if have else 0,
                                                _p(ws), _p(q), _p(s), B, H * W, C, eps, act, slope, _dt(x), _stream()), "uig_instnorm_act_fwd_mx")
            y._uig_mx = (q, s)
        elif pre is not None and pre[0].numel() == B * pre[1] * C * 2:      # statistics already accumulated by the conv epilogue
            L.check(lib.uig_instnorm_act_fwd_pre(_p(x), _p(residual), _p(y), _p(stats), _p(pre[0]), pre[1], B, H * W, C, eps, act,
                                                 slope, _dt(x), _stream()), "uig_instnorm_act_fwd_pre")
        else:
            L.check(lib.uig_instnorm_act_fwd(_p(x), _p(residual), _p(y), _p(stats), _p(ws), B, H * W, C, eps, act, slope,
                                             _dt(x), _stream()), "uig_instnorm_act_fwd")
        ctx.act, ctx.slope, ctx.has_res, ctx.skip_link = act, slope, residual is not None, skip_link
        ctx.save_for_backward(x, stats)
        if (FUSE_BWD_STATS or MIRROR_BST) and ctx.needs_input_grad[0]:
            # the convolution that consumes y produces this norm's dy in its input-gradient launch: hand it what that launch needs
            # to emit this norm's backward statistics from its epilogue (conv_dgrad / StripDesc::bst_*)
            y._uig_bst = (x, stats, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        dy = dy.contiguous()
        dx = instnorm_backward(dy, x, stats, ctx.act, ctx.slope, ctx.mx_bwd)
        dres = dy if ctx.has_res else None
        if dres is not None and ctx.skip_link is not None and ctx.needs_input_grad[1]:
            ctx.skip_link.grad, dres = dres, None       # handed to the block's first conv (SkipLink), not to autograd
        return dx, dres, None, None, None, None, None, None


def instnorm_backward(dy, x, stats, act, slope, emit_mx=False):
    """dx of InstanceNorm(+activation).  dx is also the gradient of the convolution output in front of the norm: its
    per-channel column sums are that conv's bias gradient.  The apply kernel emits them as per-block partials (no second
    pass over dx); the conv backward picks them up through the attribute `_uig_colsum` (same tensor object: the conv
    output feeds only this norm)."""
    B, H, W, C = x.shape
    lib = L.lib()
    ws = torch.empty((int(lib.uig_instnorm_workspace_floats(B, H * W, C)),), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    slabs = int(lib.uig_instnorm_bwd_colsum_slabs(B, H * W, C, _dt(x)))
    cpart = torch.empty((slabs * C * 2,), device=x.device, dtype=torch.float32)
    pre = getattr(dy, "_uig_bst_partial", None)
    gm = getattr(dy, "_uig_bst_gm", None)
    if gm is not None and tuple(gm.shape) != (B, C, 2):
        gm = None
    if FUSED_IN_BWD and gm is None and pre is None and not emit_mx and 4 * B <= _TICKET_WORDS:
        nb = int(lib.uig_instnorm_bwd_fused_applicable(B, H * W, C, _dt(x)))
        if nb > 0:      # one launch, one pass over dy and x: the blocks of an image synchronise inside the kernel
            part = torch.empty((B * nb * C * 2,), device=x.device, dtype=torch.float32)
            gmt = torch.empty((B * C * 2,), device=x.device, dtype=torch.float32)
            cp = torch.empty((B * nb * C * 2,), device=x.device, dtype=torch.float32)
            q = s = None
            if emit_mx:
                q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
                s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
            arena = ticket_arena(x.device).data_ptr()
            L.check(lib.uig_instnorm_act_bwd_fused(_p(dy), _p(x), _p(stats), _p(dx), _p(part), _p(gmt), _p(cp), arena + 2 * _TICKET_WORDS * 4,
                                                   arena + 3 * _TICKET_WORDS * 4, _p(q), _p(s), B, H * W, C, act, slope, _dt(x), _stream()),
                    "uig_instnorm_act_bwd_fused")
            if emit_mx:
                dx._uig_mx = (q, s)
            dx._uig_colsum = (cp, nb, C)
            return dx
    if gm is not None or (IN_TICKETS and pre is None and B <= _TICKET_WORDS):
        # round 4: (mean g, mean g*xhat) final out of the launch that wrote dy -> the apply launch alone; else this norm's own statistics
        # pass finalises itself (two launches instead of three)
        q = s = None
        if emit_mx:
            q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
            s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
        L.check(lib.uig_instnorm_act_bwd_colsum_t(_p(dy), _p(x), _p(stats), _p(dx), _p(ws), _p(cpart), None, 0, _p(gm),
                                                  None if gm is not None else _tickets(x.device, 1, B), _p(q), _p(s),
                                                  B, H * W, C, act, slope, _dt(x), _stream()), "uig_instnorm_act_bwd_colsum_t")
        if emit_mx:
            dx._uig_mx = (q, s)
    elif pre is not None and pre[0].numel() == B * pre[1] * C * 2:      # statistics already produced by the launch that wrote dy
        q = s = None
        if emit_mx:
            q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
            s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
        L.check(lib.uig_instnorm_act_bwd_colsum_pre(_p(dy), _p(x), _p(stats), _p(dx), _p(ws), _p(cpart), _p(pre[0]), pre[1], _p(q), _p(s),
                                                    B, H * W, C, act, slope, _dt(x), _stream()), "uig_instnorm_act_bwd_colsum_pre")
        if emit_mx:
            dx._uig_mx = (q, s)
    elif emit_mx:      # dx is the dy of an fp8 convolution: its MX form comes out of the same launch
        q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        s = torch.empty((B, H, W, C // 32), device=x.device, dtype=torch.uint8)
        L.check(lib.uig_instnorm_act_bwd_colsum_mx(_p(dy), _p(x), _p(stats), _p(dx), _p(ws), _p(cpart), _p(q), _p(s), B, H * W, C, act, slope,
                                                   _dt(x), _stream()), "uig_instnorm_act_bwd_colsum_mx")
        dx._uig_mx = (q, s)
    else:
        L.check(lib.uig_instnorm_act_bwd_colsum(_p(dy), _p(x), _p(stats), _p(dx), _p(ws), _p(cpart), B, H * W, C, act, slope,
                                                _dt(x), _stream()), "uig_instnorm_act_bwd_colsum")
    dx._uig_colsum = (cpart, slabs // B, C)
    return dx


# ----------------------------------------------------------------------------------------- conv that applies the norm in front of it
def norm_conv_applicable(c1: torch.Tensor, layers) -> bool:
    """can `layers` (one ConvLayer, or the same layer of two networks) consume the RAW convolution output c1 and apply the
    InstanceNorm(+act) in front of them themselves (NormConvFn)?  c1 must carry its norm's statistics partials (`_uig_in_partial`)."""
    pre = getattr(c1, "_uig_in_partial", None)
    if not NORM_CONV or pre is None or c1.dtype != torch.bfloat16 or not c1.is_contiguous():
        return False
    spec = layers[0].spec
    B, H, W, C = c1.shape
    if pre[0].numel() != B * pre[1] * C * 2 or any(l.fp8 or l.spec.__dict__ != spec.__dict__ for l in layers):
        return False
    if not (spec.kind == "conv" and spec.k == 3 and spec.stride == 1 and spec.pad == 1 and C == spec.cin_p == spec.cin and spec.cout_store == spec.cout):
        return False
    pm = L.PAD_REFLECT if spec.reflect else L.PAD_ZERO
    return L.lib().uig_conv3x3_innorm_applicable(B, H, W, C, spec.cout, pm, spec.cout_store, L.BF16) == 1


class _BwdShim:
    """what _conv_backward reads from an autograd ctx, for a convolution that lives inside a composite Function"""

    def __init__(self, x, needs, in_hw):
        self.saved_tensors, self.needs_input_grad, self.in_hw, self.skip_link, self.bst = (x, None), needs, in_hw, None, None


class NormConvFn(Function):
    """c2 = conv(act(InstanceNorm(c1))) with the norm applied by the convolution launch itself (uig_conv3x3_innorm_fwd): the second
    convolution of a ResBlock reads the first one's raw output and normalises its own input strip in LDS - the norm's apply pass
    (one read and one write of the whole tensor) is gone; its finalize launch (conv1's epilogue partials -> (mean, rstd)) stays.
    When a backward pass follows, the launch also stores the normalised activations h (the weight-gradient operand: a write hidden
    behind the MFMAs instead of a read + write pass).  One or two networks (paired launch).
    Backward = the convolution's ordinary backward on (h, dy) followed by the norm's ordinary backward on (c1, stats): the same
    kernels on bitwise the same tensors as the path with the apply pass."""

    @staticmethod
    def forward(ctx, c1, n_act, n_slope, n_eps, w1, b1, w2, b2, layer1, layer2, group):
        layers = (layer1,) if layer2 is None else (layer1, layer2)
        spec = layer1.spec
        B, H, W, C = c1.shape
        lib = L.lib()
        need_bwd = any(ctx.needs_input_grad)
        part1, nslab1 = c1._uig_in_partial
        stats = torch.empty((B, C, 2), device=c1.device, dtype=torch.float32)
        L.check(lib.uig_instnorm_finalize(_p(part1), nslab1, _p(stats), B, H * W, C, n_eps, _stream()), "uig_instnorm_finalize")
        h = torch.empty_like(c1) if need_bwd else None
        y = torch.empty((B, H, W, spec.cout_store), device=c1.device, dtype=c1.dtype)
        part = None
        if layer1.emit_in_stats and in_stats_fusable(spec, H, W, B, c1.dtype):
            part = torch.empty((B * (H * W // 64) * spec.cout_store * 2,), device=c1.device, dtype=torch.float32)
        L.check(lib.uig_conv3x3_innorm_fwd(_p(c1), _p(stats), n_act, n_slope, _p(h),
                                           _p(layer1.wp_fwd), _p(b1), _p(layer2.wp_fwd) if layer2 is not None else None, _p(b2), group, _p(part), _p(y),
                                           B, H, W, C, spec.cout, L.PAD_REFLECT if spec.reflect else L.PAD_ZERO, spec.cout_store, spec.act, spec.slope,
                                           _dt(c1), _stream()), "uig_conv3x3_innorm_fwd")
        if part is not None:
            y._uig_in_partial = (part, H * W // 64)
        ctx.layers, ctx.group, ctx.norm, ctx.in_hw = layers, group, (n_act, n_slope), (H, W)
        ctx.save_for_backward(c1, stats if need_bwd else None, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        c1, stats, h = ctx.saved_tensors
        layers = ctx.layers
        ng = ctx.needs_input_grad
        needs = (ng[0], ng[4], ng[5]) + ((ng[6], ng[7]) if len(layers) == 2 else ())
        dh, *grads = _conv_backward(_BwdShim(h, needs, ctx.in_hw), dy, layers, ctx.group)
        dc1 = instnorm_backward(dh.contiguous(), c1, stats, ctx.norm[0], ctx.norm[1]) if ng[0] else None
        grads = list(grads) + [None] * (4 - len(grads))
        return (dc1, None, None, None, grads[0], grads[1], grads[2], grads[3], None, None, None)


def norm_conv(c1, norm, layer1, layer2=None, group=0):
    """conv(norm(c1)) with the norm applied inside the convolution launch; `norm` = the InstNormAct module between the two
    convolutions (act, slope, eps)"""
    layer1.ensure_packed()
    if layer2 is not None:
        layer2.ensure_packed()
        return NormConvFn.apply(c1, norm.act, norm.slope, norm.eps, layer1.weight, layer1.bias, layer2.weight, layer2.bias, layer1, layer2, group)
    return NormConvFn.apply(c1, norm.act, norm.slope, norm.eps, layer1.weight, layer1.bias, None, None, layer1, None, 0)


# ----------------------------------------------------------------------------------------- fused losses
def _loss_ws(dev):
    return torch.empty((int(L.lib().uig_loss_workspace_floats()),), device=dev, dtype=torch.float32)


class L1LossFn(Function):
    """weight * mean|a - b| over the n_real unpadded elements; the gradient is produced in the same pass."""

    @staticmethod
    def forward(ctx, a, b, weight, n_real, unit_grad=False):
        ctx.unit_grad = unit_grad
        loss = torch.empty((1,), device=a.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(a) if need else None
        L.check(L.lib().uig_l1_loss_fwd_bwd(_p(a), _p(b), _p(loss), _p(grad), _p(_loss_ws(a.device)), a.numel(), n_real,
                                            weight, _dt(a), _stream()), "uig_l1_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        if ctx.unit_grad:                # the loss is a root of backward() with gradient 1: the stored gradient IS the answer
            return grad, None, None, None, None
        out = torch.empty_like(grad)
        L.check(L.lib().uig_scale_by_scalar(_p(grad), _p(g.contiguous().float()), _p(out), grad.numel(), _dt(grad), _stream()), "uig_scale_by_scalar")
        return out, None, None, None, None


class MSEConstFn(Function):
    """weight * mean((a - target)^2)  (LSGAN adversarial loss against a constant label)."""

    @staticmethod
    def forward(ctx, a, target, weight, unit_grad=False):
        ctx.unit_grad = unit_grad
        loss = torch.empty((1,), device=a.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(a) if need else None
        L.check(L.lib().uig_mse_const_fwd_bwd(_p(a), target, _p(loss), _p(grad), _p(_loss_ws(a.device)), a.numel(), weight,
                                              _dt(a), _stream()), "uig_mse_const_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        if ctx.unit_grad:
            return grad, None, None, None
        out = torch.empty_like(grad)
        L.check(L.lib().uig_scale_by_scalar(_p(grad), _p(g.contiguous().float()), _p(out), grad.numel(), _dt(grad), _stream()), "uig_scale_by_scalar")
        return out, None, None, None


def l1_loss(a, b, weight=1.0, n_real=None, unit_grad=False):
    """unit_grad=True: the caller promises the loss is passed to backward() as a root with gradient exactly 1 (the train
    step does, with an explicit ones tensor); the backward then returns the gradient computed in the forward pass as is."""
    return L1LossFn.apply(a, b, float(weight), int(a.numel() if n_real is None else n_real), bool(unit_grad))


def mse_const(a, target, weight=1.0, unit_grad=False):
    return MSEConstFn.apply(a, float(target), float(weight), bool(unit_grad))


def backward_unit(losses):
    """torch.autograd.backward(losses) with ONE shared ones tensor as every root's gradient (autograd would otherwise fill a
    fresh ones_like per root: ten 1-element kernels per train step)."""
    one = torch.ones((1,), device=losses[0].device, dtype=losses[0].dtype)
    torch.autograd.backward(losses, grad_tensors=[one] * len(losses))


def adam_flat(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    L.check(L.lib().uig_adam_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream()), "uig_adam_flat")


def new_adam_state(device, step: int = 0, lr_scale: float = 1.0) -> torch.Tensor:
    """the 16-byte device record of adam_flat_graph: {int step; float lr*scale/bc1; float 1/sqrt(bc2); float lr_scale}"""
    st = torch.zeros(4, device=device, dtype=torch.int32)
    st[0] = step
    st.view(torch.float32)[3] = lr_scale
    return st


def adam_flat_graph(p, g, m, v, lr, beta1, beta2, eps, state16, grad_scale=1.0):
    """Adam whose step counter lives on the device (state16: 4 x int32/float32), so a captured graph can replay it."""
    L.check(L.lib().uig_adam_flat_graph(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, _p(state16), grad_scale, _stream()), "uig_adam_flat_graph")
