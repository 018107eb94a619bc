"""Micro-benchmark of the conv kernels on the C4 layer shapes (HIP-event timing on the launch stream).
usage: python tools/bench_conv.py [size=128]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops

dev = torch.device("cuda")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [  # (cin, cout, spatial, kernel, stride, count-per-step fwd)
    (32, 32, S, 3, 1), (96, 32, S, 3, 1), (64, 64, S, 3, 1), (64, 64, S // 2, 3, 1), (192, 64, S // 2, 3, 1),
    (128, 128, S // 2, 3, 1), (128, 128, S // 4, 3, 1), (256, 256, S // 8, 3, 1), (512, 256, S // 8, 3, 1),
    (32, 32, S, 3, 2), (96, 32, S, 1, 1), (1, 32, S, 3, 1), (32, 1, S, 3, 1),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


for cin, cout, sp, k, s in SHAPES:
    p = 1 if k == 3 else 0
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, k, k, k), device=dev) / (cin * k ** 3) ** 0.5
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (k,) * 3, (s,) * 3, (p,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    gam, bet = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    st = ops.gn_stats(x, min(32, cin), 1e-6, gam, bet) if cin % 8 == 0 else None
    dw = torch.zeros_like(w)
    flops = 2.0 * y.numel() * cin * k ** 3
    t_f = timeit(lambda: plan.fwd(x))
    t_fp = timeit(lambda: plan.fwd(x, st, True)) if st is not None else float("nan")
    t_d = timeit(lambda: plan.dgrad(y))
    t_w = timeit(lambda: plan.wgrad(x, y, dw))
    print(f"{cin:4d}->{cout:4d} @{sp:3d}^3 k{k}s{s}: fwd {t_f*1e6:8.1f} us {flops/t_f/1e12:7.1f} TF | fwd+GN/SiLU prologue {t_fp*1e6:8.1f} us | "
          f"dgrad {t_d*1e6:8.1f} us {flops/t_d/1e12:7.1f} TF | wgrad {t_w*1e6:8.1f} us {flops/t_w/1e12:7.1f} TF", flush=True)
