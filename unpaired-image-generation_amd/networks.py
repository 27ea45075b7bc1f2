"""Generator / Discriminator modules (layer L3).

Same constructor surface and the same `state_dict` keys as the stock-torch restatement of the architecture
(`nn.Sequential` indices: '1.weight', '10.b.1.weight', ...): reflection pads, ReLU/LeakyReLU/Tanh are fused into the
neighbouring HIP kernels, and parameter-free `_Slot` placeholders keep the indices of the modules they replace.
`forward(x)` takes / returns ordinary logical (B,3,H,W) tensors; `forward_phys` works on the internal NHWC tensors.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as L
from . import ops


class _Slot(nn.Module):
    """Parameter-free placeholder for a module whose work is fused into a neighbouring kernel."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"fused: {self.what}"


class ConvLayer(nn.Module):
    """Conv2d / ConvTranspose2d (+ fused reflection pad, bias, epilogue activation) on the HIP implicit-GEMM kernel.
    Parameters keep torch's layouts (fp32 master copy); `repack()` refreshes the kernel-side operands."""

    def __init__(self, kind, cin, cout, k, stride=1, pad=0, pad_mode="zero", act=L.ACT_NONE, slope=0.0,
                 dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.spec = ops.ConvSpec(kind, cin, cout, k, stride, pad, pad_mode, act, slope)
        self.compute_dtype = dtype
        self.weight = nn.Parameter(torch.empty(self.spec.weight_shape(), device=device, dtype=torch.float32))
        self.bias = nn.Parameter(torch.zeros(cout, device=device, dtype=torch.float32))
        nn.init.normal_(self.weight, 0.0, 0.02)
        sf, sd = ops.packed_shapes(self.spec)
        # non-persistent: derived data, not part of the state_dict
        self.register_buffer("wp_fwd", torch.zeros(sf, device=device, dtype=dtype), persistent=False)
        self.register_buffer("wp_dgrad", torch.zeros(sd, device=device, dtype=dtype), persistent=False)
        self._packed_version = None
        self.fuse_grad_accum = True    # backward adds dW/db into an existing .grad in place (see ops.ConvFn.backward)
        self.emit_in_stats = False     # set by the network builders for convs that feed an InstanceNorm
        self.in_eps = 1e-5             # ... and that norm's eps (round 4: the convolution launch delivers (mean, rstd) final)
        self.fp8 = False               # enable_fp8(): forward + input gradient on the MX block-scaled fp8 kernel

    def enable_fp8(self):
        """Run this layer's forward and input gradient on the MX block-scaled fp8 MFMA kernel (BASELINE configs[4]): e4m3
        operands with one power-of-two scale per 32 channels, fp32 accumulate, bf16 activations, fp32 master weights; the
        weight gradient stays on the bf16 path.  Only 3x3 stride-1 pad-1 convs with 128-multiples of channels (the ResBlocks)."""
        s = self.spec
        if self.compute_dtype != torch.bfloat16 or not (s.kind == "conv" and s.k == 3 and s.stride == 1 and s.pad == 1
                                                        and s.cin % 128 == 0 and s.cout % 128 == 0 and s.act == L.ACT_NONE):
            raise ValueError(f"enable_fp8: unsupported layer ({self.extra_repr()}, {self.compute_dtype})")
        dev, t = self.weight.device, s.k * s.k
        for name, rows, cols in (("fwd", s.cout, s.cin), ("dgrad", s.cin, s.cout)):
            self.register_buffer("wq_" + name, torch.zeros((rows, t, cols), device=dev, dtype=torch.uint8), persistent=False)
            self.register_buffer("ws_" + name, torch.zeros((rows, t, cols // 32), device=dev, dtype=torch.uint8), persistent=False)
        self.fp8 = True
        self._packed_version = None

    def quantize_packed(self):
        """fp8 operands from the packed bf16 operands: [rows * taps][cols] matrices quantised along cols"""
        for wp, wq, ws in ((self.wp_fwd, self.wq_fwd, self.ws_fwd), (self.wp_dgrad, self.wq_dgrad, self.ws_dgrad)):
            L.check(L.lib().uig_mx_quantize(wp.data_ptr(), wq.data_ptr(), ws.data_ptr(), wp.shape[0] * wp.shape[1], wp.shape[2],
                                            L.BF16, torch.cuda.current_stream().cuda_stream), "uig_mx_quantize")

    def mx_active(self, B, H, W) -> bool:
        return self.fp8 and ops.mx_applicable(self.spec, B, H, W)

    def repack(self):
        ops.pack_weights(self.spec, self.weight.data, self.compute_dtype, self.wp_fwd, self.wp_dgrad)
        if self.fp8:
            self.quantize_packed()
        self._packed_version = self.weight._version

    def ensure_packed(self):
        if self._packed_version != self.weight._version:
            self.repack()

    def forward(self, x, skip_link=None):
        self.ensure_packed()
        return ops.ConvFn.apply(x, self.weight, self.bias, self, skip_link)

    def extra_repr(self):
        s = self.spec
        return f"{s.kind} {s.cin}->{s.cout} k{s.k} s{s.stride} p{s.pad}{'(reflect)' if s.reflect else ''} act={s.act}"


class InstNormAct(nn.Module):
    """InstanceNorm2d(affine=False, eps=1e-5) fused with ReLU / LeakyReLU and an optional residual add."""

    def __init__(self, act=L.ACT_NONE, slope=0.0, eps=1e-5):
        super().__init__()
        self.act, self.slope, self.eps = act, slope, eps
        self.mx_fwd = self.mx_bwd = False      # fp8 generators: also emit the MX fp8 form of the output / of the backward's dx

    def forward(self, x, residual=None, skip_link=None):
        if ops.INFER_FUSED_IN and not torch.is_grad_enabled() and not self.mx_fwd and ops.instnorm_infer_applicable(x):      # inference: no statistics kept, finalize fused into the apply launch
            return ops.instnorm_infer(x, residual, self.act, self.slope, self.eps)
        return ops.InstNormActFn.apply(x, residual, self.act, self.slope, self.eps, skip_link, self.mx_fwd, self.mx_bwd)


class ResBlock(nn.Module):
    """x + [ReflPad1, Conv3x3, IN, ReLU, ReflPad1, Conv3x3, IN](x); sub-keys b.1.*, b.5.* as in Appendix A."""

    def __init__(self, dim, dtype, device):
        super().__init__()
        self.b = nn.Sequential(
            _Slot("ReflectionPad2d(1) -> conv gather"),
            ConvLayer("conv", dim, dim, 3, 1, 1, "reflect", dtype=dtype, device=device),
            InstNormAct(L.ACT_RELU),
            _Slot("ReLU -> instnorm"),
            _Slot("ReflectionPad2d(1) -> conv gather"),
            ConvLayer("conv", dim, dim, 3, 1, 1, "reflect", dtype=dtype, device=device),
            InstNormAct(L.ACT_NONE),
        )

    def forward(self, x):
        # the skip path's gradient is summed into the first conv's input-gradient launch (ops.SkipLink), not by autograd
        link = ops.SkipLink() if (x.requires_grad and torch.is_grad_enabled()) else None
        c1 = self.b[1](x, skip_link=link)
        if not (self.b[2].mx_fwd or self.b[2].mx_bwd) and ops.norm_conv_applicable(c1, (self.b[5],)):
            c2 = ops.norm_conv(c1, self.b[2], self.b[5])       # conv2 normalises its own input strip: no apply pass in between
        else:
            c2 = self.b[5](self.b[2](c1))
        return self.b[6](c2, residual=x, skip_link=link)


class _PhysNet(nn.Sequential):
    in_ch = 3
    out_ch = 3
    compute_dtype = torch.bfloat16

    def forward_phys(self, xp):
        for m in self:
            xp = m(xp)
        return xp

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_ch:
            raise ValueError(f"expected (B,{self.in_ch},H,W), got {tuple(x.shape)}")
        yp = self.forward_phys(ops.ToPhysFn.apply(x, self.compute_dtype))
        if yp.shape[3] == self.out_ch:           # unpadded head (1-channel patch logits): (B,H,W,1) == (B,1,H,W)
            return yp.permute(0, 3, 1, 2).float()
        return ops.FromPhysFn.apply(yp, self.out_ch, x.dtype if x.dtype.is_floating_point else torch.float32)

    def conv_layers(self):
        return [m for m in self.modules() if isinstance(m, ConvLayer)]

    def _mark_in_producers(self):
        """convolutions directly followed by an InstanceNorm accumulate its statistics in their epilogue"""
        def walk(seq):
            mods = list(seq)
            for a, b in zip(mods, mods[1:]):
                if isinstance(a, ConvLayer) and isinstance(b, InstNormAct):
                    a.emit_in_stats, a.in_eps = True, b.eps
                if isinstance(a, ResBlock):
                    walk(a.b)
            if mods and isinstance(mods[-1], ResBlock):
                walk(mods[-1].b)
        walk(self)

    def repack(self):
        for m in self.conv_layers():
            m.repack()

    def param_index_at(self, module_idx: int) -> int:
        """number of parameters (in parameters() order) held by the modules in front of self[module_idx]"""
        return sum(len(list(m.parameters())) for m in list(self)[:module_idx])

    def stage_cut_modules(self, n_stages: int):
        """Module indices at whose INPUT the backward pass may be cut into `n_stages` stages (data-parallel gradient buckets),
        ascending.  Candidates: inputs of ResBlocks and of convolutions fed by an InstanceNorm / activation slot - the incoming
        gradient there is consumed by an InstanceNorm (or conv-epilogue activation) backward, which reads no side-channel
        attribute from it (the conv <- norm bias-gradient hand-off `_uig_colsum` never crosses such a boundary).
        Chosen greedily from the END of the network (backward order) so that every stage but the last carries about the same
        number of parameter bytes; the last stage (the first layers) takes what is left."""
        mods = list(self)
        cand = [i for i, m in enumerate(mods) if i > 0 and (isinstance(m, ResBlock) or (isinstance(m, ConvLayer) and isinstance(mods[i - 1], _Slot)))
                and self.param_index_at(i) > 0]
        if n_stages <= 1 or not cand:
            return []
        numel = [sum(p.numel() for p in m.parameters()) for m in mods]
        total = sum(numel)
        target = total / float(n_stages)
        cuts, acc = [], 0
        for i in range(len(mods) - 1, 0, -1):
            acc += numel[i]
            if i in cand and acc >= target and len(cuts) < n_stages - 1:
                cuts.append(i); acc = 0
        return sorted(cuts)


class Generator(_PhysNet):
    """ResNet generator (Appendix A): c7s1-64, d128, d256, n_blocks x R256, u128, u64, c7s1-3 + tanh.
    fp8=True: the ResBlock convolutions (88 % of the FLOPs) run forward and input gradient on the MX fp8 kernel."""

    def __init__(self, in_ch=3, out_ch=3, ngf=64, n_blocks=9, dtype=torch.bfloat16, device="cuda", fp8=False):
        kw = dict(dtype=dtype, device=device)
        mods = [_Slot("ReflectionPad2d(3) -> conv gather"), ConvLayer("conv", in_ch, ngf, 7, 1, 3, "reflect", **kw),
                InstNormAct(L.ACT_RELU), _Slot("ReLU -> instnorm"),
                ConvLayer("conv", ngf, ngf * 2, 3, 2, 1, **kw), InstNormAct(L.ACT_RELU), _Slot("ReLU -> instnorm"),
                ConvLayer("conv", ngf * 2, ngf * 4, 3, 2, 1, **kw), InstNormAct(L.ACT_RELU), _Slot("ReLU -> instnorm")]
        mods += [ResBlock(ngf * 4, dtype, device) for _ in range(n_blocks)]
        mods += [ConvLayer("convT", ngf * 4, ngf * 2, 3, 2, 1, **kw), InstNormAct(L.ACT_RELU), _Slot("ReLU -> instnorm"),
                 ConvLayer("convT", ngf * 2, ngf, 3, 2, 1, **kw), InstNormAct(L.ACT_RELU), _Slot("ReLU -> instnorm"),
                 _Slot("ReflectionPad2d(3) -> conv gather"),
                 ConvLayer("conv", ngf, out_ch, 7, 1, 3, "reflect", act=L.ACT_TANH, **kw), _Slot("Tanh -> conv epilogue")]
        super().__init__(*mods)
        self.in_ch, self.out_ch, self.compute_dtype = in_ch, out_ch, dtype
        self._mark_in_producers()
        if fp8:
            mods = list(self)
            for i, m in enumerate(mods):
                if isinstance(m, ResBlock):
                    m.b[1].enable_fp8(); m.b[5].enable_fp8()
                    # quantisation fused into the InstanceNorm launches on either side of the fp8 convolutions (ops.FUSE_MX_QUANT):
                    # forward: the norm whose output a ResBlock conv reads; backward: the norm whose dx is a ResBlock conv's dy
                    m.b[2].mx_fwd = m.b[2].mx_bwd = m.b[6].mx_bwd = ops.FUSE_MX_QUANT
                    m.b[6].mx_fwd = ops.FUSE_MX_QUANT and i + 1 < len(mods) and isinstance(mods[i + 1], ResBlock)
                    if not isinstance(mods[i - 1], ResBlock):      # the norm (+ slot) in front of the first ResBlock
                        prev = [k for k in mods[:i] if isinstance(k, InstNormAct)]
                        if prev:
                            prev[-1].mx_fwd = ops.FUSE_MX_QUANT


class Discriminator(_PhysNet):
    """70x70 PatchGAN (Appendix A), LSGAN head (no sigmoid)."""

    def __init__(self, in_ch=3, ndf=64, n_layers=3, dtype=torch.bfloat16, device="cuda"):
        kw = dict(dtype=dtype, device=device)
        mods = [ConvLayer("conv", in_ch, ndf, 4, 2, 1, act=L.ACT_LRELU, slope=0.2, **kw), _Slot("LeakyReLU -> conv epilogue")]
        nf = 1
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** n, 8)
            mods += [ConvLayer("conv", ndf * nf_prev, ndf * nf, 4, 2, 1, **kw), InstNormAct(L.ACT_LRELU, 0.2), _Slot("LeakyReLU -> instnorm")]
        nf_prev, nf = nf, min(2 ** n_layers, 8)
        mods += [ConvLayer("conv", ndf * nf_prev, ndf * nf, 4, 1, 1, **kw), InstNormAct(L.ACT_LRELU, 0.2), _Slot("LeakyReLU -> instnorm")]
        mods += [ConvLayer("conv", ndf * nf, 1, 4, 1, 1, **kw)]
        super().__init__(*mods)
        self.in_ch, self.out_ch, self.compute_dtype = in_ch, 1, dtype
        self._mark_in_producers()


def pair_forward_phys(net1: _PhysNet, net2: _PhysNet, x: torch.Tensor, taps: dict | None = None) -> torch.Tensor:
    """Run two networks of identical architecture in lockstep on one stacked batch: the first half of x goes through net1,
    the second half through net2, every convolution as ONE paired launch (ops.PairConvFn); InstanceNorm / activations are
    per-sample and simply see the whole batch.  Numerically identical to net1(x[:h]) and net2(x[h:]).
    taps: {module index: None} is filled with the INPUT tensor of those modules (cut points of the staged backward)."""
    if x.shape[0] % 2:
        raise ValueError("pair_forward_phys: the stacked batch must be even")
    g = x.shape[0] // 2

    def conv(l1, l2, t, link=None):
        if l1.spec.__dict__ != l2.spec.__dict__:
            raise ValueError("pair_forward_phys: the two networks differ")
        l1.ensure_packed(); l2.ensure_packed()
        return ops.PairConvFn.apply(t, l1.weight, l1.bias, l2.weight, l2.bias, l1, l2, g, link)

    for idx, (m1, m2) in enumerate(zip(net1, net2)):
        if taps is not None and idx in taps:
            taps[idx] = x
        if isinstance(m1, ConvLayer):
            x = conv(m1, m2, x)
        elif isinstance(m1, ResBlock):
            link = ops.SkipLink() if (x.requires_grad and torch.is_grad_enabled()) else None
            c1 = conv(m1.b[1], m2.b[1], x, link)
            if not (m1.b[2].mx_fwd or m1.b[2].mx_bwd) and ops.norm_conv_applicable(c1, (m1.b[5], m2.b[5])):
                c2 = ops.norm_conv(c1, m1.b[2], m1.b[5], m2.b[5], g)
            else:
                c2 = conv(m1.b[5], m2.b[5], m1.b[2](c1))
            x = m1.b[6](c2, residual=x, skip_link=link)
        else:                      # InstNormAct / _Slot: no parameters, per-sample
            x = m1(x)
    return x
