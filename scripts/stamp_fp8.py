"""Diagnostic: in-kernel s_memtime sums of the MX fp8 strip kernel (STAMP build, uig_debug_set_mx_stamps): per wave
{total, step wait + barrier, DMA issue (+ mirror pixels), fragment reads + MFMAs, epilogue, row table, tiles}.
s_memtime counts at 100 MHz on gfx950 (constant clock), so the sums are in 10 ns units; shares are what matters.
python scripts/stamp_fp8.py [B=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda") for _ in range(2)]
for l in ls:
    l.emit_in_stats = True; l.enable_fp8(); l.ensure_packed()
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
r = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
mxf = (ls[0].wq_fwd, ls[0].ws_fwd, ls[1].wq_fwd, ls[1].ws_fwd)
mxb = (ls[0].wq_dgrad, ls[0].ws_dgrad, ls[1].wq_dgrad, ls[1].ws_dgrad)
f = lambda: ops.conv_forward(ls[0].spec, x, ls[0].wp_fwd, ls[0].bias, pair=(ls[1].wp_fwd, ls[1].bias, B // 2), want_in_stats=True, mx=mxf)
g = lambda: ops.conv_dgrad(ls[0].spec, x, ls[0].wp_dgrad, (64, 64), pair=(ls[1].wp_dgrad, None, B // 2), res_add=r, mx=mxb)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): f()
    torch.cuda.synchronize()
def timed(fn, n=200):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
q = lambda: ops.mx_quantize(x)
tq = timed(q)
for name, fn in (("fwd + stats", f), ("reflect dgrad (mirror pixels) + res", g)):
    ts = {}
    for rep in range(2):
        for n in (8, 4):
            lib.uig_debug_set_mx_issuers(n); ts.setdefault(n, []).append(timed(fn))
    lib.uig_debug_set_mx_issuers(4)
    print(f"== {name}, B={B}: per call incl. the activation quantiser launch ({tq:.1f} us alone): " + ", ".join(f"{n} issuing waves {v[0]:.1f} / {v[1]:.1f} us" for n, v in ts.items()))
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    lib.uig_debug_set_mx_stamps(buf.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.uig_debug_set_mx_stamps(None)
    b = buf.view(256, 8, 8).double()
    b = b[b[..., 0].min(dim=1).values > 0]
    names = ["total", "step wait + barrier", "DMA issue (+ mirror pixels)", "fragment reads + MFMAs", "epilogue", "row table", "tiles"]
    tot = b[..., 0].flatten().median()
    for i, n in enumerate(names):
        col = b[..., i].flatten()
        print(f"  {n:32s} median {col.median():9.0f}  ({col.min():9.0f} .. {col.max():9.0f})  {100 * col.median() / tot:5.1f} % of total")
    rest = (b[..., 0] - b[..., 1:6].sum(-1)).flatten()
    print(f"  {'unaccounted (prologue, between)':32s} median {rest.median():9.0f}")
