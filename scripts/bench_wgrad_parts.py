"""ResBlock weight gradient: partial-GEMM launch and reduce launch timed separately, image-row kernel vs generic kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib; lib = L.lib()
dt = torch.bfloat16
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
s = torch.cuda.current_stream().cuda_stream
for B in (4, 8):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    out = torch.zeros(256, 256, 3, 3, device="cuda")
    for rows in (0, 1):
        lib.uig_debug_set_wgrad_rows(rows)
        splits = int(lib.uig_wgrad_splits(B, 64, 64, 256, 64, 64, 256, 3, 3, 1, 1, L.BF16, 512))
        ws = torch.empty(splits * 256 * 9 * 256, device="cuda", dtype=torch.float32)
        part = lambda: L.check(lib.uig_wgrad_partial(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), B, 64, 64, 256, 64, 64, 256, 3, 3, 1, 1, L.PAD_REFLECT, splits, L.BF16, s), "p")
        red = lambda: L.check(lib.uig_wgrad_reduce(ws.data_ptr(), out.data_ptr(), 256, 256, 9, splits, 256, 256, 1, s), "r")
        gf = 2 * 256 * 2304 * B * 4096 / 1e9
        tp, tr = t(part), t(red)
        print(f"B{B} {'rows   ' if rows else 'generic'} splits={splits:3d}: partial {tp:6.1f} us ({gf / tp / 1e3:6.1f} TF) | reduce {tr:5.1f} us ({splits * 2.36 / tr * 1e3 / 1e3:5.2f} TB/s) | sum {tp + tr:6.1f}")
lib.uig_debug_set_wgrad_rows(1)
