"""Diagnostic (needs a -DUIG_X_STAMP build of wgrad_rows.hip via UIG_LIB_PATH): in-kernel clock and loop / epilogue split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
L = u.lib; lib = L.lib()
dt = torch.bfloat16
s = torch.cuda.current_stream().cuda_stream
for B in (4, 8):
    x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
    splits = int(lib.uig_wgrad_splits(B, 64, 64, 256, 64, 64, 256, 3, 3, 1, 1, L.BF16, 512))
    ws = torch.empty(splits * 256 * 9 * 256, device="cuda", dtype=torch.float32)
    part = lambda: L.check(lib.uig_wgrad_partial(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), B, 64, 64, 256, 64, 64, 256, 3, 3, 1, 1, L.PAD_REFLECT, splits, L.BF16, s), "p")
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(200): part()
        torch.cuda.synchronize()
    st = ws.view(splits, -1)[:, :6].cpu()
    cyc_loop, ns_loop, cyc_epi, ns_epi = [float(st[:, i].median()) for i in range(4)]
    print(f"B{B}: loop {cyc_loop:.0f} cycles = {ns_loop / 100:.2f} us -> clock {cyc_loop / (ns_loop * 10) :.2f} GHz ; per K-step {cyc_loop / (B * 64 / splits):.0f} cycles | epilogue {cyc_epi:.0f} cycles = {ns_epi / 100:.2f} us | in-loop waits per K-step: dma {float(st[:, 4].median()) / (B * 64 / splits):.0f}, barrier {float(st[:, 5].median()) / (B * 64 / splits):.0f} cycles")
