#!/bin/bash
# Diagnostic: shader clock and package power (rocm-smi) while the dominant strip-conv launch runs back to back (variant = $1: 0 default, 9 lean phased).
cd "$(dirname "$0")/.."
python - "$1" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib(); lib.uig_debug_set_strip_pk(int(sys.argv[1]), 0)
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
x = (torch.rand(16, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    for _ in range(500): ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, 8), want_in_stats=True)
    torch.cuda.synchronize(); n += 500
print(f"variant {sys.argv[1]}: {n} launches in {time.time() - t0:.1f} s = {(time.time() - t0) / n * 1e6:.1f} us per launch (back to back, incl. launch gaps)")
PY
PID=$!
sleep 8
for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" ; sleep 1; done
wait $PID
