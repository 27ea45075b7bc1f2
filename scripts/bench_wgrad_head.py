"""7x7 64->3 output-conv weight gradient at 256x256: partial launch and reduce launch timed separately, head kernel vs generic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
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
    x = (torch.rand(B, 256, 256, 64, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(B, 256, 256, 8, device="cuda") * 2 - 1).to(dt)
    out = torch.zeros(3, 64, 7, 7, device="cuda")
    for head in (0, 1):
        lib.uig_debug_set_wgrad_head(head)
        splits = int(lib.uig_wgrad_splits(B, 256, 256, 8, 256, 256, 64, 7, 7, 1, 3, L.BF16, 512))
        ws = torch.empty(splits * 8 * 49 * 64, device="cuda", dtype=torch.float32)
        part = lambda: L.check(lib.uig_wgrad_partial(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), B, 256, 256, 8, 256, 256, 64, 7, 7, 1, 3, L.PAD_REFLECT, splits, L.BF16, s), "p")
        red = lambda: L.check(lib.uig_wgrad_reduce(ws.data_ptr(), out.data_ptr(), 8, 64, 49, splits, 3, 64, 1, s), "r")
        print(f"B{B} {'head   ' if head else 'generic'} splits={splits:3d}: partial {t(part):6.1f} us | reduce {t(red):5.1f} us")
lib.uig_debug_set_wgrad_head(1)
