"""bench.py -- train-step voxels/sec of the 3-D DDPM U-Net at 128^3 bf16 on N MI355X (BASELINE.json metric).

One JSON line on rank 0 (contract in the task statement).  A "step" = noise -> q-sample -> UNet forward -> MSE ->
backward -> (gradient all-reduce) -> clip_grad_norm_ -> AdamW on one synthetic batch already resident in HBM.
  python bench.py [--gpus N --steps K --warmup W]
N > 1: one rank per GPU over RCCL -- either started by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / WORLD_SIZE in the environment) or, without a launcher, by bench.py itself as a child process.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# C4 of SURVEY 8d / BASELINE.json configs[3]; configs[1] (C2) is the same net at 96^3, batch 2
C4 = dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64, 128, 256),
          attention_levels=(False, False, False, True), num_head_channels=(0, 0, 0, 64), norm_num_groups=32,
          strides=[[1] * 3] + [[2] * 3] * 3, kernel_sizes=[[3] * 3] * 4, paddings=[[1] * 3] * 4)
MFMA_PEAK_BF16 = 2.5e15  # dense, MI355X_MICROARCH.md "Chip-level parameters"


def synthetic_volume(shape, seed, device):
    """uniform [0,1) inside a centred ellipsoid, 0 outside (NIfTI-like intensity volume)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(shape, generator=g)
    grids = torch.meshgrid(*[torch.linspace(-1, 1, s) for s in shape[2:]], indexing="ij")
    return (x * (sum(v ** 2 for v in grids) <= 0.9)).to(device)


def kernel_source_hash():
    """sha256 over the conv kernel sources: ties a PMC measurement to the binary it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv.hip", "conv27.hip", "conv27_kernel.h", "conv_common.h", "common.h"):
        with open(os.path.join(ROOT, "medical_image_generation_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kind, size):
    """HBM bytes per launch of the roofline kernels from the PMC passes committed under profiles/ (tools/pmc_traffic.py: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/pmc_conv.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    streaming reads on gfx950).  Only a measurement of THESE kernel sources at THIS shape counts: anything else reports null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("source_hash") == kernel_source_hash() and d.get("size") == size and kind in d.get("hbm_bytes_per_launch", {}):
            best = d["hbm_bytes_per_launch"][kind]
    return best


def kernel_roofline(kind, size, iters=20):
    """One conv kernel on its most frequent shape in the step (k3 s1 32->32 at full resolution: 7 launches each of forward,
    dgrad and wgrad per step; the weight-gradient kernels are the largest kernel family of the step by total time).  At this shape the
    library runs the rolling-halo variants: k_conv27r<0> (forward) and k_conv_wgrad3 (weight gradient).
    HIP-event timing on the launch stream; algorithmic flops = 2 * voxels * 32 * 32 * 27."""
    from medical_image_generation_amd import hipops as ops
    dev = torch.device("cuda")
    x = torch.randn((1, size, size, size, 32), device=dev).to(torch.bfloat16)
    w = torch.randn((32, 32, 3, 3, 3), device=dev) / 30
    plan = ops.ConvPlan(1, (size,) * 3, 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan.pack(w)
    dy = plan.fwd(x)
    dw = torch.zeros_like(w)
    fn = (lambda: plan.fwd(x)) if kind == "fwd" else (lambda: plan.wgrad(x, dy, dw))
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / iters
    flops = 2.0 * size ** 3 * 32 * 32 * 27
    name = {"fwd": "k_conv27r<0>", "wgrad": "k_conv_wgrad3 (+ k_wgrad_reduce, ~5 us, inside the timed launch pair)"}[kind]
    return {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
            "frac": flops / sec / MFMA_PEAK_BF16, "traffic": pmc_traffic(kind, size),
            "kernel": f"{name}: k3 s1 32->32 @{size}^3", "avg_launch_us": sec * 1e6,
            "algorithmic_bytes": 2.0 * size ** 3 * 32 * 2,  # fwd: read x + write y; wgrad: read x + read dy (bf16, 32 ch)
            "hbm_GBps_algorithmic": 2.0 * size ** 3 * 32 * 2 / sec / 1e9}


def cpu_baseline(size=128, timed=2):
    """The oracle (CPU restatement of the reference, fp32) timed on this host on the SAME workload as the GPU line: full train steps
    (q-sample, forward, MSE, backward, clip, AdamW) of the C4 U-Net on one size^3 volume, 1 warm-up + `timed` timed steps, median
    (BASELINE.md section 3).  Threads = this job's CPU share (16 per GPU on the pool), not the host's 256.  About a minute."""
    from oracle import nets, step
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    net = nets.DiffusionModelUNet(**C4)
    for n, p in net.named_parameters():  # un-zero the zero_module'd convs so the backward is not trivially sparse
        if float(p.detach().abs().max()) == 0:
            torch.nn.init.normal_(p, std=0.02)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-5)
    sched = step.DDPMSchedule()
    x0 = torch.rand(1, 1, size, size, size)
    times = []
    for k in range(1 + timed):
        t0 = time.perf_counter()
        step.ddpm_train_step(net, opt, sched, x0, torch.randn_like(x0), torch.tensor([(137 * k + 500) % 1000]))
        times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2] if timed > 1 else times[-1]
    return {"value": size ** 3 / dt, "unit": "voxels/s", "cores": threads, "kind": "port",
            "sample": f"1 warm-up + {timed} timed full train steps (fwd+bwd+clip+AdamW) of the C4 U-Net on one {size}^3 volume (the GPU "
                      f"line's workload), fp32; median {dt:.1f} s/step (warm-up {times[0]:.1f} s)"}


def step_breakdown(tr, x0, noise, t):
    """GPU milliseconds of ONE eager step by C-ABI entry point (HIP events around every library call on the launch stream): which
    kernel families the step spends its time in, so the whole-step MFMA fraction is explained by the record itself."""
    from medical_image_generation_amd import _lib
    with _lib.profile_calls() as prof:
        tr.step(x0, noise, t)
    torch.cuda.synchronize()
    by = prof.summary()
    groups = {"conv k3 fwd": ("mi_conv_fwd",), "conv dgrad": ("mi_conv_dgrad",), "conv wgrad": ("mi_conv_wgrad",),
              "groupnorm": ("mi_gn_",), "attention": ("mi_attn_", "mi_softmax_", "mi_linear_wgrad"), "gemm / transpose": ("mi_gemm_", "mi_transpose_"),
              "weight pack": ("mi_conv_pack",), "optimizer": ("mi_adam", "mi_sumsq", "mi_axpy"),
              "resample / copy / add": ("mi_upsample", "mi_space", "mi_depth", "mi_copy", "mi_add_bf16", "mi_avgpool")}
    out, used = {}, set()
    for gname, prefixes in groups.items():
        ms = sum(v for k, v in by.items() if k.startswith(prefixes))
        used |= {k for k in by if k.startswith(prefixes)}
        if ms > 0:
            out[gname] = round(ms, 3)
    out["other"] = round(sum(v for k, v in by.items() if k not in used), 3)
    out["sum"] = round(sum(by.values()), 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="per-GPU batch")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only time the dominant kernels (for rocprofv3 cross-checks)")
    args = ap.parse_args()

    if args.roofline_only:
        torch.cuda.set_device(0)
        print(json.dumps({"roofline": kernel_roofline("wgrad", args.size), "roofline_fwd": kernel_roofline("fwd", args.size)}), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (torch.distributed.run) before this
        # process has touched the GPU, and hand its exit code on.  Never exec / re-exec from a process that has initialised HIP.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("MI_DIST_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo" only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    torch.manual_seed(42)  # identical initial weights on every rank
    net = DiffusionModelUNet(**C4)
    for n, p in net.named_parameters():
        if float(p.detach().abs().max()) == 0:  # zero_module'd tensors: randomise (otherwise half the backward sees zeros)
            torch.nn.init.normal_(p, std=0.02)
    net = net.to(dev)
    tr = DDPMTrainer(net, lr=2e-5, optimizer="AdamW", max_grad_norm=1.0, device=dev)
    shape = (args.batch, 1, args.size, args.size, args.size)
    x0 = synthetic_volume(shape, 42 + rank, dev)
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    noise = torch.empty(shape, device=dev)
    t = torch.empty(args.batch, dtype=torch.int64, device=dev)

    use_graph = not args.no_graph
    noise.normal_(generator=gen), t.random_(0, 1000, generator=gen)
    if use_graph:
        try:
            tr.capture(x0, noise, t)
            _, noise, t = tr._static  # the graphs read these buffers: fresh noise / timesteps are drawn INTO them each step
        except Exception as exc:  # capture is an optimisation: never let it take the run down
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr, flush=True)
            use_graph = False

    def one_step(use_graph):
        noise.normal_(generator=gen)
        t.random_(0, 1000, generator=gen)
        return tr.step_graph() if use_graph else tr.step(x0, noise, t)

    for _ in range(args.warmup):
        one_step(use_graph)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step(use_graph)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
        ok = torch.tensor([1.0 if use_graph else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        use_graph = bool(ok.item())
    voxels = world * args.batch * args.size ** 3 * args.steps
    # A step that computes garbage is not a measurement (and NaN operands even run FASTER: less MFMA power, higher clocks).
    if not bool(torch.isfinite(loss.detach().float()).all()) or not bool(torch.isfinite(tr.arena.data).all()):
        raise RuntimeError(f"bench: non-finite loss / parameters after the timed steps (loss = {float(loss)}): refusing to report a number")

    if rank == 0:
        # algorithmic flops of one step of this model (engine counters: 2*MACs, bwd = dgrad + wgrad)
        from medical_image_generation_amd import engine as E
        c = E.Ctx(tr.arena, net._plans, grad_enabled=True)
        net._run(c, torch.zeros((args.batch, args.size, args.size, args.size, 1), dtype=torch.bfloat16, device=dev), t, need_dx=False)
        c.tape.fns.clear()
        step_flops = c.flops_fwd + c.flops_bwd
        del c
        roof = kernel_roofline("wgrad", args.size)       # largest kernel of the step by total time
        roof_fwd = kernel_roofline("fwd", args.size)     # second largest (forward and dgrad share it)
        out = {
            "metric": "3D DDPM U-Net train-step voxels/sec at 128^3 bf16", "value": voxels / dt, "unit": "voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"C4: DiffusionModelUNet num_channels=(32,64,128,256), attention at 16^3 (heads of 64), "
                                   f"{args.size}^3 x batch {args.batch}/GPU, eps-prediction, AdamW, clip 1.0",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "hipgraph": use_graph},
            "model_flops_per_step": step_flops,
            "step_mfma_frac": step_flops / (dt / args.steps) / MFMA_PEAK_BF16,
            "loss": float(loss),
            "roofline": roof,
            "roofline_fwd": roof_fwd,
        }
        if world == 1:
            out["step_breakdown_ms"] = step_breakdown(tr, x0, noise, t)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 is still timing the roofline kernels when the others get here: tear the group down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
