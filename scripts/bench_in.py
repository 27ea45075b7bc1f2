"""InstanceNorm kernel timing at the step's hot shapes (rotating buffers: HBM-realistic, no cache residency).
  python scripts/bench_in.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unpaired_image_generation_amd as u
from unpaired_image_generation_amd import ops
L = u.lib
lib = L.lib()


def ev(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(B, HW, C, tag):
    dt = torch.bfloat16
    # rotate over several buffer sets so the working set behaves like the step's (no artificial L2/MALL residency)
    NS = 6
    xs = [torch.randn(B, HW, C, device="cuda").to(dt) for _ in range(NS)]
    dys = [torch.randn(B, HW, C, device="cuda").to(dt) for _ in range(NS)]
    outs = [torch.empty_like(xs[0]) for _ in range(NS)]
    stats = torch.zeros(B * C * 2, device="cuda"); ws = torch.zeros(int(lib.uig_instnorm_workspace_floats(B, HW, C)), device="cuda")
    cpart = torch.zeros(B * 128 * C * 2, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    k = [0]
    def fwd():
        i = k[0] = (k[0] + 1) % NS
        L.check(lib.uig_instnorm_act_fwd(xs[i].data_ptr(), None, outs[i].data_ptr(), stats.data_ptr(), ws.data_ptr(), B, HW, C, 1e-5, 1, 0.0, 1, st), "f")
    def fwd_pre():
        i = k[0] = (k[0] + 1) % NS
        L.check(lib.uig_instnorm_act_fwd_pre(xs[i].data_ptr(), dys[i].data_ptr(), outs[i].data_ptr(), stats.data_ptr(), ws.data_ptr(), 16, B, HW, C, 1e-5, 0, 0.0, 1, st), "fp")
    def bwd():
        i = k[0] = (k[0] + 1) % NS
        L.check(lib.uig_instnorm_act_bwd_colsum(dys[i].data_ptr(), xs[i].data_ptr(), stats.data_ptr(), outs[i].data_ptr(), ws.data_ptr(), cpart.data_ptr(), B, HW, C, 1, 0.0, 1, st), "b")
    fwd(); torch.cuda.synchronize()
    print(f"{tag}: fwd(stats+fin+apply) {ev(fwd):7.1f} us   fwd_pre(fin+apply+res) {ev(fwd_pre):7.1f} us   bwd(stats+fin+apply+colsum) {ev(bwd):7.1f} us", flush=True)


if True:
    run(16, 64 * 64, 256, "  B16 64x64 C256 ")
    run(16, 128 * 128, 128, "  B16 128x128 C128")
    run(16, 256 * 256, 64, "  B16 256x256 C64 ")
