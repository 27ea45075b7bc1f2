"""Dominant kernel (ResBlock 3x3 256->256 at 64x64, paired launch) on the three 256x128 strip variants, interleaved rounds in
one process (rule 24): one-tile-per-block (round 1), persistent with DMA issue at the top of the K-step, persistent with the
DMA issue spread between MFMA groups.  Forward (fused IN statistics) and reflect dgrad (+border, +residual add), batch 16 and 8;
outputs compared bitwise with the one-tile-per-block kernel (same accumulation order).  python scripts/bench_strip_pk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops, networks
lib = u.lib.lib()
dt = torch.bfloat16
l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l1.repack()
l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dt, device="cuda"); l2.repack()
def t(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
VARIANTS = {"tile/block": (3, 0, 0), "pk": (1, 0, 0), "pk no-xprefetch": (1, 12, 0), "pk nolgk": (1, 4, 0), "pk rd0-dma": (1, 8, 0), "pk rdall-dma": (1, 9, 0), "pk setprio": (1, 10, 0),
            "pk 4iss": (1, 20, 0), "pk pkrt": (1, 21, 0), "pk 8iss": (1, 23, 0),
            "pk r3": (1, 5, 0)}      # round 4: "pk" = the phased schedule (two wave groups one barrier apart, three weight stages, counted DMA waits); "pk r3": round 3's default      # round 3: "pk" issues its DMAs from four waves; 20 / 21: packed row table with four / eight issuing waves; 23: round 2's eight
if os.environ.get("PK_VARIANTS"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k == "tile/block" or k in os.environ["PK_VARIANTS"].split(",")}
lib.uig_debug_set_mirror(0)      # the input gradient in its border-buffer form on every variant, so that dx is comparable bitwise
if len(sys.argv) > 1:
    VARIANTS.update({f"pk rot g{g}": (1, 4, int(g)) for g in sys.argv[1:]})
def select(v):
    m, dm, g = VARIANTS[v]
    lib.uig_debug_set_strip(m); lib.uig_debug_set_strip_pk(dm, g)
ref, res = {}, {}
for rnd in range(3):
    for v in VARIANTS:
        select(v)
        for B in (16, 8):
            x = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B)) * 2 - 1).to(dt)
            r = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B + 1)) * 2 - 1).to(dt)
            f = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=(l2.wp_fwd, l2.bias, B // 2), want_in_stats=True)
            g = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
            y, dx = f(), g()
            if v == "tile/block" and B not in ref: ref[B] = (y.clone(), y._uig_in_partial[0].clone(), dx.clone())
            eq = (torch.equal(y, ref[B][0]), torch.equal(y._uig_in_partial[0], ref[B][1]), torch.equal(dx, ref[B][2]))
            tf, tg = t(f), t(g)
            res.setdefault((v, B), []).append((tf, tg))
            fl = 2.0 * B * 4096 * 256 * 2304 / 1e6
            print(f"round {rnd} {v:12s} B{B:2d}: fwd {tf:6.1f} us ({fl/tf:5.0f} TF)  dgrad+border+res {tg:6.1f} us ({fl/tg:5.0f} TF) | bitwise == tile/block: y {eq[0]} stats {eq[1]} dx {eq[2]}", flush=True)
            assert all(eq), "persistent kernel differs from the one-tile-per-block kernel"
print("\nmedians (us):")
for (v, B), xs in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    fs, gs = sorted(a for a, _ in xs), sorted(b for _, b in xs)
    fl = 2.0 * B * 4096 * 256 * 2304 / 1e6
    print(f"  B{B:2d} {v:12s} fwd {fs[len(fs)//2]:6.1f} ({fl/fs[len(fs)//2]:5.0f} TF = {fl/fs[len(fs)//2]/2500:.3f} of peak)   dgrad {gs[len(gs)//2]:6.1f} ({fl/gs[len(gs)//2]:5.0f} TF)")
lib.uig_debug_set_strip(1); lib.uig_debug_set_strip_pk(0, 0); lib.uig_debug_set_mirror(1)

# round 3: the mirror-pixel input gradient (the step's form) on the issuing-wave variants, bitwise against the default persistent kernel
lib.uig_debug_set_mirror(1)
mres, mref = {}, {}
for rnd in range(3):
    for v in [k for k in ("pk", "pk r3", "pk 4iss", "pk pkrt", "pk 8iss") if k in VARIANTS]:
        select(v)
        for B in (16, 8):
            x = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B)) * 2 - 1).to(dt)
            r = (torch.rand(B, 64, 64, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(B + 1)) * 2 - 1).to(dt)
            g = lambda: ops.conv_dgrad(l1.spec, x, l1.wp_dgrad, (64, 64), pair=(l2.wp_dgrad, None, B // 2), res_add=r)
            dx = g()
            if v == "pk" and B not in mref: mref[B] = dx.clone()
            assert torch.equal(dx, mref[B]), f"mirror-pixel dgrad of {v} differs from the default persistent kernel"
            mres.setdefault((v, B), []).append(t(g))
select("pk")
print("mirror-pixel dgrad + res, medians (us):")
for (v, B), xs in sorted(mres.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"  B{B:2d} {v:12s} {sorted(xs)[len(xs)//2]:6.1f}")
