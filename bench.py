#!/usr/bin/env python
"""bench.py — images/sec of the full CycleGAN train step (3x256x256) on N MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full optimisation step on a per-GPU batch of `--batch` (default 4: BASELINE.json
configs[1]) synthetic (real_A, real_B) pairs; value = global pairs / second (weak scaling).  Rank 0 prints ONE JSON line
with the contract fields plus `roofline` (dominant kernel: the 256->256 3x3 ResBlock convolution, HIP-event timed
here), `g_fwd` (the paired generator forward of the step as its own HIP graph: the north_star's MFMA-utilisation target)
and, at N=1, `cpu_baseline` (the CPU oracle's train step on this host's cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_G256 = 99.10e9      # generator forward FLOPs / image @256^2 (hook-counted on the oracle)
F_D256 = 6.29e9       # discriminator forward FLOPs / image
PEAK_BF16 = 2.5e15    # dense bf16 MFMA peak (MI355X_MICROARCH.md: Chip-level parameters)
PEAK_F32 = 157.3e12
PEAK_FP8 = 5.0e15     # dense MX-scaled fp8 MFMA peak


def step_flops(size, n_blocks=9):
    s = (size / 256.0) ** 2
    return (18 * F_G256 + 16 * F_D256) * s if n_blocks == 9 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="pairs per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: BASELINE configs[4] (use --batch 8): ResBlock convs fwd + dgrad on MX block-scaled fp8 MFMA, bf16 elsewhere")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-iters", type=int, default=300, help="timed launches / 2 of the dominant kernel (3x this many back-to-back, the first third is warm-up)")
    ap.add_argument("--force-comm", action="store_true", help="run the RCCL exchange even at world size 1 (plumbing test)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.force_comm:
        assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import unpaired_image_generation_amd as u
    assert u.lib.lib().uig_device_ok() == 1, "bench.py needs an MI355X (gfx950)"
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    torch.manual_seed(0)                                   # identical replicas on every rank
    model = u.CycleGAN(n_blocks=9, dtype=dtype, device=dev, use_graph=not args.no_graph, force_exchange=args.force_comm, fp8=args.dtype == "fp8")
    model.broadcast_params(0)
    torch.manual_seed(1000 + rank)                         # different data shard per rank
    B, S = args.batch, args.size
    real_A = torch.rand(B, 3, S, S, device=dev) * 2 - 1
    real_B = torch.rand(B, 3, S, S, device=dev) * 2 - 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        model.train_step(real_A, real_B, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step(real_A, real_B, sync=False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.train_step(real_A, real_B)              # also proves the step still produces finite losses
    assert all(v == v for v in losses.values()), losses

    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    sf = step_flops(S)
    out = {
        "metric": f"images/sec full CycleGAN train step, 3x{args.size}x{args.size}", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"CycleGAN train step: 9-block G_A/G_B + 70x70 PatchGAN D_A/D_B, {S}x{S}, "
                               f"batch {B}/GPU, " + ("MX block-scaled fp8 (e4m3, one E8M0 scale per 32 channels) ResBlock convs fwd+dgrad, bf16 elsewhere"
                                                     if args.dtype == "fp8" else f"{args.dtype} MFMA conv path") + ", fp32 master weights + Adam",
                   "global_batch": world * B, "image": f"3x{S}x{S}", "parallelism": f"dp{world}",
                   "hip_graph": model.graph_active},
        "step_tflops": round(sf * B / (ms * 1e-3) / 1e12, 2),
        "step_mfma_frac": round(sf * B / (ms * 1e-3) / (PEAK_F32 if args.dtype == "f32" else PEAK_BF16), 4),
        "losses": {k: round(v, 4) for k, v in losses.items()},
    }
    if rank == 0:
        out["roofline"] = dominant_kernel_roofline(u, torch, dev, dtype, 4 * B, S // 4, args.kernel_iters, fp8=args.dtype == "fp8")
        out["g_fwd"] = generator_forward_mfma(u, torch, model, dev, dtype, B, S)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, S, B)
        print(json.dumps(out), flush=True)
    # ordered teardown owned by the product (CycleGAN.close: drain, drop graphs / packers / streams, drain), then the
    # process group, then a normal interpreter exit
    model.close()
    del model
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()


def generator_forward_mfma(u, torch, model, dev, dtype, B, S):
    """MFMA utilisation of the 9-block generator FORWARD exactly as the step runs it: G_A || G_B in lockstep over the 4B-image
    stack [real_B; real_A | real_B; real_A] (fake + identity passes of both generators, paired launches), captured into one HIP
    graph and replayed back to back between HIP events on the launch stream.  99.10 GFLOP per image at 256^2."""
    from unpaired_image_generation_amd.networks import pair_forward_phys
    x = (torch.rand(4 * B, S, S, 8, device=dev) * 2 - 1).to(dtype)
    x[..., 3:] = 0
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            pair_forward_phys(model.G_A, model.G_B, x)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = pair_forward_phys(model.G_A, model.G_B, x)
    for _ in range(3):
        g.replay()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 4 * B * F_G256 * (S / 256.0) ** 2
    peak = PEAK_BF16 if dtype == torch.bfloat16 else PEAK_F32
    assert bool(torch.isfinite(y.float()).all())
    del g
    return {"what": f"paired G_A||G_B forward over {4 * B} images {S}x{S} (the step's fake+identity pass), one HIP graph",
            "ms": round(ms, 4), "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_frac": round(flops / (ms * 1e-3) / peak, 4),
            "flops": flops, "target_mfma_frac": 0.40}


def dominant_kernel_roofline(u, torch, dev, dtype, nimg, hw, iters, fp8=False):
    """The ResBlock 3x3 reflect-pad conv (256->256 on hw x hw) exactly as the step launches it most often: ONE paired launch
    over the 4B-image stack [G_A: real_A, real_B | G_B: real_B, real_A] (two weight sets).  88 % of generator FLOPs.
    HIP events (torch.cuda.Event on the launch stream) around `iters` back-to-back launches on random data."""
    from unpaired_image_generation_amd import ops, networks
    l1 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dtype, device=dev)
    l2 = networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dtype, device=dev)
    if fp8:
        l1.enable_fp8(); l2.enable_fp8()
    l1.repack(); l2.repack()
    x = (torch.rand(nimg, hw, hw, 256, device=dev) * 2 - 1).to(dtype)
    pair = (l2.wp_fwd, l2.bias, nimg // 2)
    if fp8:      # the fp8 kernel alone, on pre-quantised operands (the quantisation pass is a separate, HBM-bound kernel)
        xq, xs = ops.mx_quantize(x)
        y = torch.empty_like(x)
        mx = (l1.wq_fwd, l1.ws_fwd, l2.wq_fwd, l2.ws_fwd)
        launch = lambda: ops._conv3x3_mx(xq, xs, mx, l1.bias, l2.bias, nimg // 2, y, 256, u.lib.PAD_REFLECT, u.lib.GATHER_DIRECT, u.lib.ACT_NONE, 0.0)
    else:
        launch = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=pair)
    # back-to-back launches; the first third is warm-up (the clock the chip settles at under this load is what counts:
    # five warm-up launches after the host-side pause that follows the training loop read 20 % slow), the rest is timed
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3 * iters):
        if i == iters:
            e0.record()
        launch()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (2 * iters)
    flops = 2.0 * nimg * hw * hw * 256 * 2304
    peak = PEAK_FP8 if fp8 else (PEAK_BF16 if dtype == torch.bfloat16 else PEAK_F32)
    ach = flops / (us * 1e-6)
    # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs,
    # gfx950 FETCH_SIZE x2 correction: scripts/prof_dominant.sh + scripts/pmc_summary.py); null when the shape differs.
    # NOT measured in this run (PMC counters need rocprofv3): quoted from the newest committed profile of the same shape, with its file name
    traffic_from_profile = None
    try:
        pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_dominant_pmc.json"))
        if pmc and dtype == torch.bfloat16 and not fp8:
            j = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
            if j.get("images") == nimg and j.get("hw") == hw:
                traffic_from_profile = {"hbm_bytes_per_launch": j.get("hbm_bytes_per_launch"), "file": "profiles/" + pmc[-1],
                                        "kernel": j.get("kernel")}
    except OSError:
        pass
    eb = 1 if fp8 else (2 if dtype == torch.bfloat16 else 4)      # operand bytes per element (the output is bf16 on the fp8 path)
    alg = nimg * hw * hw * 256 * (eb + (2 if fp8 else eb)) + 2 * 256 * 2304 * eb
    kid = u.lib.lib().uig_debug_last_conv_kernel() if not fp8 else -1
    kname = {-1: "conv_strip_fp8_kernel<448> (MX e4m3 v_mfma_scale_f32_16x16x128, 256x128 tiles, persistent blocks) [%s operands]",u.lib.K_STRIP_PK: "conv_strip_pk_kernel<%s,448> (256x128 tiles, persistent blocks)", u.lib.K_STRIP256: "conv_strip_kernel<%s,256,128>"}.get(kid, "kernel id %d <%%s>" % kid)
    return {"kernel": (kname + " conv3x3 256->256 reflect, paired G_A|G_B launch (ResBlock fwd)") % ("fp8" if fp8 else "bf16" if dtype == torch.bfloat16 else "f32"),
            "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None, "traffic_from_profile": traffic_from_profile, "algorithmic_bytes": alg,
            "avg_us": round(us, 2), "gemm": f"M={nimg * hw * hw} N=256 K=2304 (2 weight sets)", "flops_per_launch": flops}


def cpu_baseline(torch, size, batch):
    """The CPU oracle's train step (stock torch fp32) on this host's cores at the benchmark's own batch: a bounded sample (about 10-30 s).
    Threads = the cores this process may actually use (affinity mask, capped at 16 = the GPU box's CPU share per GPU);
    os.cpu_count() over-reports inside a cgroup and oversubscribing oneDNN makes the step several times slower."""
    from oracle.torch_oracle import CycleGANOracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    o = CycleGANOracle(n_blocks=9)
    rA = torch.rand(batch, 3, size, size) * 2 - 1
    rB = torch.rand(batch, 3, size, size) * 2 - 1
    print(f"[bench] cpu_baseline: oracle train step on {cores} threads ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    o.train_step(rA, rB)                       # first step (includes oneDNN primitive creation)
    first = time.perf_counter() - t0
    print(f"[bench] cpu_baseline: first step {first:.1f} s", file=sys.stderr, flush=True)
    n, dt = 0, first
    if first < 12.0:                           # keep the whole sample within ~30 s
        n = 2 if first < 6.0 else 1
        t0 = time.perf_counter()
        for _ in range(n):
            o.train_step(rA, rB)
        dt = (time.perf_counter() - t0) / n
    return {"value": round(batch / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (stock torch {torch.__version__} fp32) full train step, B={batch} {size}x{size}: "
                      + (f"1 warm-up + {n} timed steps" if n else "the first step only (it took > 12 s)")}


if __name__ == "__main__":
    main()
