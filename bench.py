#!/usr/bin/env python
"""bench.py — images/sec of the full CycleGAN train step (3x256x256) on N MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1: either  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
       or plain  python bench.py --gpus N ...  : without WORLD_SIZE in the environment the process starts that launcher itself as a
       CHILD (before torch is imported or the GPU touched: no exec of a GPU process) and exits with its status.

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


def exchange_bandwidth(torch, dist, model, dev, world, iters=10):
    """time the all-reduces one step issues (same bucket slices of the flat gradient buffers, same process group) without the compute"""
    bufs, saved = [], []
    for grad, buckets in ((model.grp_G.grad, model.buckets_G), (model.grp_D.grad, model.buckets_D)):
        saved.append((grad, grad.clone()))
        bufs += [(grad, a, b) for a, b in buckets]
    nbytes = sum((b - a) * 4 for _, a, b in bufs)
    st = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(st):
        for it in range(iters + 2):
            if it == 2:
                e0.record(st)
            for grad, a, b in bufs:
                dist.all_reduce(grad[a:b], op=dist.ReduceOp.SUM)
        e1.record(st)
    e1.synchronize()
    for grad, keep in saved:
        grad.copy_(keep)                                       # the sums of the timing loop are not gradients
    ms = e0.elapsed_time(e1) / iters
    t = torch.tensor([ms], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())
    alg = nbytes / (ms * 1e-3) / 1e9
    return {"what": "the step's gradient all-reduces (bucketed slices of the flat fp32 G and D gradient buffers) alone, back to back, max over ranks",
            "buckets": len(bufs), "bytes_per_step": nbytes, "allreduce_ms": round(ms, 4), "alg_bw_GBps": round(alg, 2),
            "bus_bw_GBps": round(alg * 2 * (world - 1) / world, 2), "world": world}


def step_flops(size, n_blocks=9):
    s = (size / 256.0) ** 2
    return (18 * F_G256 + 16 * F_D256) * s if n_blocks == 9 else None


def step_time_floor(flops, size, dtype):
    """seconds the step's FLOPs take at the dense MFMA peak of the dtype each of them runs in"""
    if dtype == "f32":
        return flops / PEAK_F32
    if dtype == "bf16":
        return flops / PEAK_BF16
    share8 = (18 * F_G256 * 0.878 * 2.0 / 3.0) / (18 * F_G256 + 16 * F_D256)      # ResBlock conv fwd + dgrad of all 18 generator pass-equivalents
    return flops * share8 / PEAK_FP8 + flops * (1.0 - share8) / PEAK_BF16


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1, a free port) as a child process with this script and the same arguments, and return its exit status.  The parent
    has not imported torch and never touches the GPU; the ranks are fresh processes."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_launch():
    """--dry-launch: the launch plumbing alone, no GPU - every rank joins a gloo group, the ranks are counted with an all-reduce
    and rank 0 prints ONE line (tests/test_dp_cpu.py runs this with --gpus 2 in the CPU container)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"dry_launch": True, "world": world, "ranks_seen": seen, "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}), flush=True)


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
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configs[k-1] as a preset: 1 = G6@64 forward parity smoke (no timing), 2 = the headline (256^2, batch 4, bf16), "
                         "3 = config 2 per GPU on --gpus ranks (batch 32 over 8), 4 = 512^2 batch 2 per GPU, 5 = 256^2 batch 8 per GPU, MX fp8 ResBlock convs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short timings of configs[3] / configs[4] appended to the default line")
    ap.add_argument("--dry-launch", action="store_true", help="exercise the rank launch only (gloo, no GPU): rank 0 prints one line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))       # nothing below has run: torch is not imported, the GPU is untouched
    if args.dry_launch:
        assert int(os.environ.get("WORLD_SIZE", "1")) == args.gpus, "WORLD_SIZE does not match --gpus"
        return dry_launch()
    if args.config is not None:
        if args.config == 1:
            import __graft_entry__
            __graft_entry__.smoke()            # configs[0] is the CPU-reference plumbing case: parity of the G6@64 forward, nothing to time
            return
        preset = {2: dict(batch=4, size=256, dtype="bf16"), 3: dict(batch=4, size=256, dtype="bf16"),
                  4: dict(batch=2, size=512, dtype="bf16"), 5: dict(batch=8, size=256, dtype="fp8")}[args.config]
        for k, v in preset.items():
            setattr(args, k, v)

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
        # fraction of the MFMA time floor: FLOPs / peak of the dtype they run in.  fp8: the ResBlock convs' forward and input
        # gradient (0.878 of the generator FLOPs x 2 of its 3 GEMMs; generators = 18 F_G of the step) at the fp8 peak, the rest bf16
        "step_mfma_frac": round(step_time_floor(sf * B, S, args.dtype) / (ms * 1e-3), 4),
        "step_mfma_peak": {"f32": "157.3 TF f32 MFMA", "bf16": "2.5 PF dense bf16", "fp8": "FLOP-weighted: ResBlock conv fwd+dgrad at 5 PF MX-fp8, the rest at 2.5 PF bf16"}[args.dtype],
        "losses": {k: round(v, 4) for k, v in losses.items()},
    }
    if world > 1 or args.force_comm:
        # the step's gradient exchange alone (every rank takes part): the buckets the staged backward all-reduces, back to back on one
        # stream, HIP-event timed; bus bandwidth = 2 (N - 1) / N x bytes / time; at N = 1 (plumbing run) the factor is 0
        out["comm"] = exchange_bandwidth(torch, dist, model, dev, world)
    if rank == 0:
        out["roofline"] = dominant_kernel_roofline(u, torch, dev, dtype, 4 * B, S // 4, args.kernel_iters, fp8=args.dtype == "fp8")
        if args.dtype == "bf16" and S % 4 == 0:
            fam = strip_family_in_step(u, torch, dev, dtype, B, S // 4)
            out["roofline"]["in_step_frac"] = fam["frac"]
            out["strip_family"] = fam
        out["g_fwd"] = generator_forward_mfma(u, torch, model, dev, dtype, B, S)
        out["g_fwd"]["breakdown_ms"] = generator_forward_breakdown(u, torch, model, dev, dtype, B, S)
    # ordered teardown owned by the product (CycleGAN.close: drain, drop graphs / packers / streams, drain) before the other
    # configurations build their own models
    model.close()
    del model
    if rank == 0:
        if world == 1 and args.config is None and not args.no_other_configs and (B, S, args.dtype) == (4, 256, "bf16") and not args.no_graph:
            out["other_configs"] = other_configs(u, torch, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, S, B)
        print(json.dumps(out), flush=True)
    # then the process group, then a normal interpreter exit
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush(); sys.stderr.flush()


def other_configs(u, torch, dev):
    """Short timings of the other single-GPU BASELINE configurations so that the driver's default run sees them: configs[3]'s
    per-GPU shape (512x512, batch 2, bf16) and configs[4]'s (256x256, batch 8, MX fp8 ResBlock convs) - each a fresh model, HIP
    graphs, 5 warm-up + 10 timed full train steps (wall clock around a synchronised loop, as the headline)."""
    res = {}
    for name, (B, S, dt) in (("512_b2_bf16", (2, 512, "bf16")), ("256_b8_fp8", (8, 256, "fp8")), ("256_b8_bf16", (8, 256, "bf16"))):
        torch.manual_seed(0)
        m = u.CycleGAN(n_blocks=9, dtype=torch.bfloat16, device=dev, use_graph=True, fp8=dt == "fp8")
        rA = torch.rand(B, 3, S, S, device=dev) * 2 - 1
        rB = torch.rand(B, 3, S, S, device=dev) * 2 - 1
        for _ in range(5):
            m.train_step(rA, rB, sync=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            m.train_step(rA, rB, sync=False)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 10 * 1e3
        losses = m.train_step(rA, rB)
        sf = step_flops(S) * B
        res[name] = {"workload": f"full train step, {S}x{S}, batch {B}, {dt}", "ms_per_step": round(ms, 3), "images_per_s": round(B / (ms * 1e-3), 2),
                     "steps": 10, "warmup": 5, "hip_graph": m.graph_active, "finite": all(v == v for v in losses.values()),
                     "step_mfma_frac": round(step_time_floor(sf, S, dt) / (ms * 1e-3), 4)}
        m.close()
        del m
    return res


def strip_family_in_step(u, torch, dev, dtype, B, hw):
    """The strip-convolution launches of ONE train step as their own HIP graph, replayed back to back between HIP events: per
    ResBlock conv pair (18 of them) the step runs a forward launch over 4B images and one over 2B images (both emit the
    InstanceNorm statistics of the following norm) and the two reflect-pad input-gradient launches (mirror pixels; every
    second one also sums the ResBlock's skip gradient) - 72 launches.  `frac` = their FLOPs / the graph's time / peak: the
    dominant kernel AS THE STEP USES IT (small one-tile-per-block launches and the input gradient included), where `roofline.frac`
    is its most frequent launch alone.  Still without the cold caches the step's InstanceNorm kernels leave behind: the
    rocprofv3 step profile under profiles/ carries that number."""
    from unpaired_image_generation_amd import ops, networks
    ls = [networks.ConvLayer("conv", 256, 256, 3, 1, 1, "reflect", dtype=dtype, device=dev) for _ in range(2)]
    for l in ls:
        l.repack()
    x16 = (torch.rand(4 * B, hw, hw, 256, device=dev) * 2 - 1).to(dtype)
    x8 = x16[:2 * B].contiguous()
    spec = ls[0].spec

    def launches():
        for x in (x16, x8):
            n = x.shape[0]
            for i in range(18):
                y = ops.conv_forward(spec, x, ls[0].wp_fwd, ls[0].bias, pair=(ls[1].wp_fwd, ls[1].bias, n // 2), want_in_stats=True)
                ops.conv_dgrad(spec, y, ls[0].wp_dgrad, (hw, hw), (ls[1].wp_dgrad, None, n // 2), x if i % 2 == 0 else None)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        launches()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
        launches()
    for _ in range(5):
        g.replay()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * (4 * B + 2 * B) * hw * hw * 256 * 2304 * 18 * 2
    peak = PEAK_BF16 if dtype == torch.bfloat16 else PEAK_F32
    del g
    return {"what": f"the 72 strip-conv launches of one step (18 x [fwd+stats over {4 * B} and {2 * B} images, reflect-pad input gradient of each, "
                    "every second one with the skip gradient]) as one HIP graph", "ms": round(ms, 4), "avg_us_per_launch": round(ms * 1e3 / 72, 2),
            "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "frac": round(flops / (ms * 1e-3) / peak, 4), "flops": flops}


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


def generator_forward_breakdown(u, torch, model, dev, dtype, B, S, reps=5):
    """Where the paired generator forward's time goes, by operator family: the same 4B-image pass launched eagerly with a HIP event
    in front of and behind every convolution launch and every InstanceNorm (statistics finalize + apply), averaged over `reps` passes.
    Eager launches leave gaps the graph replay does not have, so the families are reported as measured AND the graph time beside
    them (`g_fwd.ms`); their ratios are what the breakdown is for."""
    from unpaired_image_generation_amd import ops, networks
    from unpaired_image_generation_amd.networks import pair_forward_phys
    x = (torch.rand(4 * B, S, S, 8, device=dev) * 2 - 1).to(dtype)
    x[..., 3:] = 0
    marks = []                                              # (family, start event, end event)
    real_conv, real_norm = ops.conv_forward, networks.InstNormAct.forward

    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e

    def conv_forward(spec, *a, **k):
        fam = "stem_head_7x7" if spec.k == 7 else ("resblock_conv3x3" if (spec.kind == "conv" and spec.stride == 1) else "stride2_down_up")
        e0 = ev(); y = real_conv(spec, *a, **k); marks.append((fam, e0, ev()))
        return y

    def norm_forward(self_, xx, residual=None, skip_link=None):
        e0 = ev(); y = real_norm(self_, xx, residual, skip_link); marks.append(("instnorm", e0, ev()))
        return y

    ops.conv_forward, networks.InstNormAct.forward = conv_forward, norm_forward
    try:
        with torch.no_grad():
            pair_forward_phys(model.G_A, model.G_B, x)      # warm
            marks.clear()
            t0 = ev()
            for _ in range(reps):
                pair_forward_phys(model.G_A, model.G_B, x)
            t1 = ev()
        torch.cuda.synchronize(dev)
    finally:
        ops.conv_forward, networks.InstNormAct.forward = real_conv, real_norm
    out = {}
    for fam, e0, e1 in marks:
        out[fam] = out.get(fam, 0.0) + e0.elapsed_time(e1) / reps
    out = {k: round(v, 4) for k, v in out.items()}
    out["eager_total"] = round(t0.elapsed_time(t1) / reps, 4)
    return out


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
        # with the InstanceNorm statistics of the following norm requested, as every forward launch of the step does
        launch = lambda: ops.conv_forward(l1.spec, x, l1.wp_fwd, l1.bias, pair=pair, want_in_stats=True)
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
    kname = {-1: "conv_strip_fp8_kernel<448> (MX e4m3 v_mfma_scale_f32_16x16x128, 256x128 tiles, persistent blocks) [%s operands]",u.lib.K_STRIP_PK: "conv_strip_pk_kernel<%s,440,phased> (256x128 tiles, persistent blocks, two wave groups one barrier apart)", u.lib.K_STRIP256: "conv_strip_kernel<%s,256,128>"}.get(kid, "kernel id %d <%%s>" % kid)
    return {"kernel": (kname + " conv3x3 256->256 reflect, paired G_A|G_B launch (ResBlock fwd, emitting the InstanceNorm statistics as in the step)") % ("fp8" if fp8 else "bf16" if dtype == torch.bfloat16 else "f32"),
            "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None, "traffic_from_profile": traffic_from_profile, "algorithmic_bytes": alg,
            "avg_us": round(us, 2), "gemm": f"M={nimg * hw * hw} N=256 K=2304 (2 weight sets)", "flops_per_launch": flops,
            "clock_note": "peak = nominal 2.5 PFLOP/s at 2.4 GHz; this launch loops at the package power limit (rocm-smi 1350-1360 W of 1400) and the part "
                          "lowers the shader clock to 2.07-2.24 GHz under it; register-resident bf16 MFMAs on random operands sustain 0.81 of nominal "
                          "(scripts/probes/mfma_power.hip); in-kernel stamps: 1182-1190 cycles per K-step for 1024 of MFMA (DESIGN.md 5.000)"}


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
