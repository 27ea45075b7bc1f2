"""What does finalising the InstanceNorm statistics INSIDE the apply kernel cost, as a function of the number of partials per
image?  (finalize launch + apply launch) against the one-launch form (uig_instnorm_act_fwd_infer) at the ResBlock norm's shape,
rotating buffers, np = 8 / 16 / 32 / 64 partials per image.  A kernel launch here has a floor of ~4.7 us (rocprofv3: one-thread
kernels), so the one-launch form pays whenever its prologue costs less than that.   python scripts/bench_in_fin.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unpaired_image_generation_amd as u
L = u.lib
lib = L.lib()


def ev(fn, iters=60, warm=10):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(6): fn()
    for _ in range(warm): g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters / 6 * 1e3


def run(B, HW, C):
    dt = torch.bfloat16
    NS = 6
    xs = [torch.randn(B, HW, C, device="cuda").to(dt) for _ in range(NS)]
    rs = [torch.randn(B, HW, C, device="cuda").to(dt) for _ in range(NS)]
    outs = [torch.empty_like(xs[0]) for _ in range(NS)]
    stats = torch.zeros(B * C * 2, device="cuda")
    k = [0]
    for np_ in (8, 16, 32, 64):
        part = torch.rand(B * np_ * C * 2, device="cuda")
        part[1::2] += 10.0      # sum of squares > (sum)^2 / n
        def pre():
            i = k[0] = (k[0] + 1) % NS
            L.check(lib.uig_instnorm_act_fwd_pre(xs[i].data_ptr(), rs[i].data_ptr(), outs[i].data_ptr(), stats.data_ptr(), part.data_ptr(), np_, B, HW, C, 1e-5, 0, 0.0, 1,
                                                 torch.cuda.current_stream().cuda_stream), "pre")
        def fin():
            i = k[0] = (k[0] + 1) % NS
            L.check(lib.uig_instnorm_act_fwd_infer(xs[i].data_ptr(), rs[i].data_ptr(), outs[i].data_ptr(), part.data_ptr(), np_, None, B, HW, C, 1e-5, 0, 0.0, 1,
                                                   torch.cuda.current_stream().cuda_stream), "fin")
        print(f"B{B} HW{HW} C{C} np={np_:3d}: finalize + apply {ev(pre):6.1f} us   apply with in-block finalize {ev(fin):6.1f} us", flush=True)


run(16, 64 * 64, 256)
run(8, 64 * 64, 256)
run(1, 64 * 64, 256)
