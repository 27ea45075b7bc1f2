"""Launch the dominant kernel (ResBlock 3x3 reflect conv 256->256 on 64x64, 16 images = the paired G_A|G_B launch of the
batch-4 step) N times on random data; used under rocprofv3 (--kernel-trace --stats, and separate --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stats = len(sys.argv) > 2 and sys.argv[2] == "stats"      # also emit the following InstanceNorm's statistics, as every forward launch of the step does
fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"          # the MX fp8 kernel at configs[4]'s launch size (32 images: batch 8 per GPU, paired)
dt = torch.bfloat16
layer = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda")
layer2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda")
layer.repack(); layer2.repack()
if fp8:
    layer.enable_fp8(); layer2.enable_fp8(); layer.repack(); layer2.repack()
    x = (torch.rand(32, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    xq, xs = ops.mx_quantize(x)
    y = torch.empty_like(x)
    mx = (layer.wq_fwd, layer.ws_fwd, layer2.wq_fwd, layer2.ws_fwd)
    for _ in range(n):
        ops._conv3x3_mx(xq, xs, mx, layer.bias, layer2.bias, 16, y, 256, u.lib.PAD_REFLECT, u.lib.GATHER_DIRECT, u.lib.ACT_NONE, 0.0)
    torch.cuda.synchronize()
    print("done fp8", float(y.float().abs().mean()))
    sys.exit(0)
x = (torch.rand(16, 64, 64, 256, device="cuda") * 2 - 1).to(dt)     # the paired 4B-image launch of the batch-4 step
for _ in range(n):
    y = ops.conv_forward(layer.spec, x, layer.wp_fwd, layer.bias, pair=(layer2.wp_fwd, layer2.bias, 8), want_in_stats=stats)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
