import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
def check(kind, cin, cout, k, s, p, pm, B, H, W, direct=True):
    ops.REFLECT_DGRAD_DIRECT = direct
    torch.manual_seed(1)
    layer = networks.ConvLayer(kind, cin, cout, k, s, p, pm, dtype=torch.float32, device="cuda")
    w = layer.weight.detach().cpu() * 3
    with torch.no_grad(): layer.weight.copy_(w)
    x = torch.rand(B, cin, H, W) * 2 - 1
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    xin = F.pad(xr, (p, p, p, p), mode="reflect") if pm == "reflect" else xr
    yr = F.conv2d(xin, wr, layer.bias.detach().cpu(), s, 0 if pm == "reflect" else p)
    dy = torch.randn_like(yr); yr.backward(dy)
    xp = ops.to_nhwc(x.cuda(), torch.float32).requires_grad_(True)
    yp = layer(xp); yp.backward(ops.to_nhwc(dy.cuda(), torch.float32, yp.shape[3]))
    dx = ops.from_nhwc(xp.grad, cin).cpu()
    d = (dx - xr.grad).abs()
    bad = (d > 1e-4 * xr.grad.abs().max()).nonzero()
    print(f"{kind} {cin}->{cout} k{k} {pm} B{B} {H}x{W} direct={direct}: dx Linf {float(d.max()):.2e} (max {float(xr.grad.abs().max()):.2e}) bad {len(bad)}", end="")
    if len(bad):
        hs = sorted(set(bad[:, 2].tolist())); ws = sorted(set(bad[:, 3].tolist())); bs = sorted(set(bad[:, 0].tolist()))
        print(f"  imgs {bs[:6]} rows {hs[:12]} cols {ws[:12]} chans {sorted(set(bad[:,1].tolist()))[:8]}")
    else: print()
    dW = layer.weight.grad.cpu(); print(f"      dW rel {float((dW-wr.grad).norm()/wr.grad.norm()):.2e}")
check("conv", 256, 256, 3, 1, 1, "reflect", 1, 16, 16)
check("conv", 256, 256, 3, 1, 1, "reflect", 1, 16, 16, direct=False)
check("conv", 256, 256, 3, 1, 1, "reflect", 1, 8, 8)
check("conv", 256, 256, 3, 1, 1, "reflect", 2, 32, 32)
check("conv", 256, 256, 3, 1, 1, "reflect", 2, 32, 32, direct=False)
check("conv", 64, 3, 7, 1, 3, "reflect", 1, 64, 64)
check("conv", 64, 3, 7, 1, 3, "reflect", 1, 32, 32)
check("conv", 64, 3, 7, 1, 3, "reflect", 1, 72, 104)
check("conv", 3, 64, 7, 1, 3, "reflect", 1, 64, 64)
