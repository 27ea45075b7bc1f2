import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
from oracle.torch_oracle import Generator as OG, init_weights
def run(tag, par=True, direct=True, instats=True, colsum=True, strip=1):
    ops.PARALLEL_BACKWARD, ops.REFLECT_DGRAD_DIRECT = par, direct
    u.lib.lib().uig_debug_set_strip(strip)
    B, H, W = 1, 64, 64
    torch.manual_seed(100 + H + W)
    og = init_weights(OG(n_blocks=2)); g = u.Generator(n_blocks=2, dtype=torch.float32); g.load_state_dict(og.state_dict())
    if not instats:
        for l in g.conv_layers(): l.emit_in_stats = False
    if not colsum:
        orig = ops.instnorm_backward
        def nb(dy, x, st, act, sl):
            dx = orig(dy, x, st, act, sl); del dx._uig_colsum; return dx
        ops.instnorm_backward = nb
    x = torch.rand(B, 3, H, W) * 2 - 1
    xr = x.clone().requires_grad_(True); yr = og(xr); t = torch.randn_like(yr); (yr * t).sum().backward()
    xg = x.cuda().requires_grad_(True); y = g(xg); (y * t.cuda()).sum().backward()
    ref = dict(og.named_parameters())
    errs = {k: float((p.grad.cpu() - ref[k].grad).norm() / (ref[k].grad.norm() + 1e-30)) for k, p in g.named_parameters() if k.endswith(".weight")}
    print(f"{tag:28s} dx rel {float((xg.grad.cpu()-xr.grad).norm()/xr.grad.norm()):.2e}  " + " ".join(f"{k.replace('.weight','')}={v:.1e}" for k, v in errs.items()), flush=True)
    if not colsum: ops.instnorm_backward = orig
run("default")
run("no parallel backward", par=False)
run("no direct dgrad", direct=False)
run("no fused IN stats", instats=False)
run("no colsum attr", colsum=False)
run("no strip kernel", strip=0)
run("all off", par=False, direct=False, instats=False, colsum=False, strip=0)
