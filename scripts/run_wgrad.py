import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); layer.repack()
x = (torch.rand(8, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
dy = (torch.rand(8, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
for _ in range(n):
    dW = ops.conv_wgrad(layer.spec, x, dy)
torch.cuda.synchronize()
print("done", float(dW.abs().mean()))
