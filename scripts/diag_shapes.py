import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from oracle.torch_oracle import Generator as OG, init_weights
for (B, H, W) in [(1, 72, 104), (3, 128, 128), (1, 32, 32), (1, 64, 64)]:
    torch.manual_seed(100 + H + W)
    og = init_weights(OG(n_blocks=2)); g = u.Generator(n_blocks=2, dtype=torch.float32); g.load_state_dict(og.state_dict())
    x = torch.rand(B, 3, H, W) * 2 - 1
    xr = x.clone().requires_grad_(True); yr = og(xr); t = torch.randn_like(yr); (yr * t).sum().backward()
    xg = x.cuda().requires_grad_(True); y = g(xg); (y * t.cuda()).sum().backward()
    d = (xg.grad.cpu() - xr.grad)
    print(f"[{B},{H},{W}] fwd Linf {float((y.detach().cpu()-yr.detach()).abs().max()):.2e}  dx Linf {float(d.abs().max()):.3e} of max {float(xr.grad.abs().max()):.3e}  rel L2 {float(d.norm()/xr.grad.norm()):.2e}")
    ref = dict(og.named_parameters())
    for k, p in g.named_parameters():
        r = ref[k].grad
        if k.endswith(".weight"):
            print(f"    {k:16s} rel L2 {float((p.grad.cpu()-r).norm()/(r.norm()+1e-30)):.2e}")
