"""Weight gradient of the stride-2 3x3 layers with 128-multiples of channels (down2: Conv2d 128 -> 256 on 128 x 128, up1: ConvTranspose2d
256 -> 128 on 64 x 64) as the step launches it - both generator passes (16 + 8 images, two networks) in ONE partial launch + its reduce -
on the stride-2 image-row kernel (round 3) against the generic split-K kernel, alternated in one process.  python scripts/bench_wgrad_s2.py [B=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
L = u.lib
lib = L.lib()
dt = torch.bfloat16
Bb = int(sys.argv[1]) if len(sys.argv) > 1 else 4

def ev(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n

for name, kind, cin, cout, hw in (("down2", "conv", 128, 256, 128), ("up1", "convT", 256, 128, 64)):
    layer = networks.ConvLayer(kind, cin, cout, 3, 2, 1, "zero", dtype=dt, device="cuda")
    spec = layer.spec
    B1, g1, B2, g2 = 4 * Bb, 2 * Bb, 2 * Bb, Bb
    Ho, Wo = spec.out_hw(hw, hw)
    xs = [(torch.rand(B, hw, hw, spec.cin_p, device="cuda") * 2 - 1).to(dt) for B in (B1, B2)]
    dys = [(torch.randn(B, Ho, Wo, spec.cout_p, device="cuda") * 0.5).to(dt) for B in (B1, B2)]
    P1, Q1, Mh, Mw, Np, Hq, Wq, Cq, pm, D0, D1 = ops._wgrad_operands(spec, xs[0], dys[0])
    P2, Q2 = ops._wgrad_operands(spec, xs[1], dys[1])[:2]
    args = (Mh, Mw, Np, Hq, Wq, Cq, 3, 3, 2, 1)
    outs = [torch.zeros(spec.weight_shape(), device="cuda") for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for rnd in range(3):
        for on in (0, 1):
            lib.uig_debug_set_wgrad_rows_s2(on)
            splits = int(lib.uig_wgrad_pair2_splits(B1, g1, B2, g2, 1, *args, L.BF16))
            per = splits * Np * 9 * Cq
            ws = torch.empty((2 * per,), device="cuda", dtype=torch.float32)
            def part():
                L.check(lib.uig_wgrad_partial_pair2(P1.data_ptr(), Q1.data_ptr(), P2.data_ptr(), Q2.data_ptr(), ws.data_ptr(), B1, g1, B2, g2, 1, *args, pm, splits, L.BF16, s), "pair2")
            def red():
                L.check(lib.uig_wgrad_reduce_pair(ws.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), Np, Cq, 9, splits, D0, D1, 0, None, None, 0, 0, 0, 0, None, None, 0, s), "reduce")
            def both():
                part(); red()
            res.setdefault(on, []).append((ev(part), ev(red), ev(both), splits))
    lib.uig_debug_set_wgrad_rows_s2(1)
    fl = 2.0 * (B1 + B2) * Mh * Mw * Np * Cq * 9 / 1e6
    for on in (0, 1):
        p, r, b, sp = sorted(res[on])[1]
        print(f"{name} ({B1}+{B2} images): {'stride-2 row kernel' if on else 'generic split-K    '}: partial {p:6.1f} us ({fl / p:5.0f} TF)  reduce {r:5.1f} us  both {b:6.1f} us  splits {sp}", flush=True)
