import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib
def check(B, C, H, W, act, res=False, scale=1.0, shift=0.0):
    torch.manual_seed(3)
    x = torch.randn(B, C, H, W) * scale + shift
    r = torch.randn(B, C, H, W)
    xr = x.clone().requires_grad_(True); rr = r.clone().requires_grad_(True)
    y = F.instance_norm(xr, eps=1e-5)
    y = F.relu(y) if act == L.ACT_RELU else y
    if res: y = y + rr
    dy = torch.randn_like(y); y.backward(dy)
    xp = ops.to_nhwc(x.cuda(), torch.float32).requires_grad_(True)
    rp = ops.to_nhwc(r.cuda(), torch.float32).requires_grad_(True) if res else None
    yp = networks.InstNormAct(act)(xp, rp); yp.backward(ops.to_nhwc(dy.cuda(), torch.float32))
    e_f = float((ops.from_nhwc(yp.detach(), C).cpu() - y.detach()).abs().max())
    d = ops.from_nhwc(xp.grad, C).cpu() - xr.grad
    print(f"IN B{B} C{C} {H}x{W} act{act} res{res} scale{scale} shift{shift}: fwd Linf {e_f:.2e}  dx rel L2 {float(d.norm()/xr.grad.norm()):.2e} Linf {float(d.abs().max()):.2e} (max {float(xr.grad.abs().max()):.2e})")
check(1, 256, 16, 16, L.ACT_RELU)
check(3, 64, 128, 128, L.ACT_RELU)
check(1, 256, 16, 16, L.ACT_RELU, scale=0.05)
check(1, 256, 16, 16, L.ACT_RELU, scale=0.05, shift=0.02)
check(1, 256, 16, 16, L.ACT_NONE, res=True, scale=0.05)
check(3, 64, 128, 128, L.ACT_RELU, scale=0.02, shift=0.05)
