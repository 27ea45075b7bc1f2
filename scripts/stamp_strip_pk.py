"""Diagnostic: in-kernel s_memtime stamps of the persistent strip kernel (STAMP build): per wave
[entry, K start tile 0, K end 0, epilogue issued 0, K start 1, K end 1, epilogue issued 1, stores drained].
python scripts/stamp_strip_pk.py [B=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
x = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
r = (torch.rand(B, 64, 64, 256, device="cuda") * 2 - 1).to(dt)
f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
g = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
z1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "zero", dtype=dt, device="cuda"); z1.repack()
z2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "zero", dtype=dt, device="cuda"); z2.repack()
gz = lambda: ops.conv_dgrad(z1.spec, x, z1.wp_dgrad, (64, 64), pair=(z2.wp_dgrad, None, B // 2))
gm = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2))
def gm7():
    lib.uig_debug_set_mirror(7); gm(); lib.uig_debug_set_mirror(1)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(100): f()
    torch.cuda.synchronize()
GRID = int(os.environ.get("STAMP_GRID", "0"))
if GRID:
    lib.uig_debug_set_strip_pk(0, GRID)      # fewer persistent blocks: does the epilogue get faster when fewer CUs store at once?
for name, fn in (("fwd+stats", f), ("reflect dgrad (mirror pixels) + res", g), ("reflect dgrad (mirror pixels)", gm), ("reflect dgrad, mirror kernel WITHOUT the fixes (diag)", gm7), ("zero-pad dgrad (plain kernel)", gz)):
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    lib.uig_debug_set_strip_stamps(buf.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.uig_debug_set_strip_stamps(None)
    b = buf.view(256, 8, 8).double()
    b = b[b[..., 0].min(dim=1).values > 0]          # blocks that ran
    d = b[..., 1:] - b[..., :-1]
    names = ["prologue (zero rows, first DMAs, row table)", "K loop tile 0", "wait + barrier behind the K loop", "epilogue: accumulators -> LDS scratch",
             "epilogue: row reads, adds, stores issued", "between tiles / drain", "(tile 1) K loop", "(tile 1) wait + barrier"]
    print(f"== {name}, B={B}: per-wave cycles, median over waves (min..max)")
    for i, n in enumerate(names[:d.shape[-1]]):
        col = d[..., i].flatten()
        print(f"  {n:48s} {col.median():9.0f}  ({col.min():9.0f} .. {col.max():9.0f})")
    tot = (b[..., 7] - b[..., 0]).flatten()
    print(f"  first 8 stamps span {tot.median():.0f} cycles")
