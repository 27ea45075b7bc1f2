import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
torch.manual_seed(0)
layer = networks.ConvLayer("conv", 3, 64, 7, 1, 3, "reflect", dtype=torch.bfloat16, device="cuda"); layer.repack()
x = torch.zeros(1, 16, 16, 8, device="cuda", dtype=torch.bfloat16); x[..., :3] = (torch.rand(1, 16, 16, 3, device="cuda") * 2 - 1).to(torch.bfloat16)
lib.uig_debug_set_cin8(0); a = ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias).float()
lib.uig_debug_set_cin8(1); b = ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias).float()
torch.cuda.synchronize()
d = (a - b).abs()
print("max diff", float(d.max()), "ref max", float(a.abs().max()))
print("per-row max diff:", [round(float(d[0, i].max()), 3) for i in range(16)])
print("per-col max diff:", [round(float(d[0, :, j].max()), 3) for j in range(16)])
print("per-channel max diff (first 16):", [round(float(d[..., c].max()), 3) for c in range(16)])
print("a[0,5,5,:4]", a[0, 5, 5, :4].tolist(), "b", b[0, 5, 5, :4].tolist())
for col in (8, 9, 10):
    print("col", col, "b:", [round(v, 3) for v in b[0, 5, col, :6].tolist()], "a same col:", [round(v, 3) for v in a[0, 5, col, :6].tolist()], "a col-8:", [round(v, 3) for v in a[0, 5, col - 8, :6].tolist()])
# which reference column does b's column 8 match best?
for col in (8, 10, 12, 14):
    errs = [(float((b[0, 5, col] - a[0, 5, j]).abs().max()), j) for j in range(16)]
    print("b col", col, "closest a col:", min(errs))
