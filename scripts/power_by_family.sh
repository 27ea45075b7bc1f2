#!/bin/bash
# Diagnostic: package power and shader clock (rocm-smi) while ONE kernel family of the train step loops for ~9 s each:
# strip conv forward / mirror-pixel input gradient, image-row weight gradient, InstanceNorm forward (statistics + apply) / backward, Adam.
cd "$(dirname "$0")/.."
for fam in strip_fwd strip_dgrad wgrad in_fwd in_bwd idle; do
python - $fam <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
fam = sys.argv[1]
L = u.lib; dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
B = 16
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
dy = (torch.randn(B, 64, 64, 256, device="cuda") * 0.5).to(dt)
if fam == "strip_fwd": f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
elif fam == "strip_dgrad": f = lambda: ops.conv_dgrad(l1.spec, dy, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=x)
elif fam == "wgrad": f = lambda: ops.conv_wgrad(l1.spec, x, dy)
elif fam == "in_fwd": f = lambda: ops.InstNormActFn.apply(x, None, L.ACT_RELU, 0.0, 1e-5)
elif fam == "in_bwd":
    y = ops.InstNormActFn.apply(x.clone().requires_grad_(True), None, L.ACT_RELU, 0.0, 1e-5)
    stats = torch.stack([x.float().mean((1, 2)), 1.0 / torch.sqrt(x.float().var((1, 2), unbiased=False) + 1e-5)], -1).contiguous()
    f = lambda: ops.instnorm_backward(dy, x, stats, L.ACT_RELU, 0.0)
else: f = None
t0 = time.time(); n = 0
while time.time() - t0 < 9:
    if f is None: time.sleep(0.2); continue
    for _ in range(300): f()
    torch.cuda.synchronize(); n += 300
print(f"{fam}: {(time.time() - t0) / max(n, 1) * 1e6:.1f} us per call (back to back)" if n else f"{fam}: no work")
PY
PID=$!
sleep 7
for i in 1 2; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*: //' | tr '\n' ' '; echo " <- $fam"; sleep 0.6; done
wait $PID
done
