"""Micro-benchmark of the factor-2 layers of the C4 / C3a nets: Upsample + conv and the k3 s2 conv, forward / data gradient / weight gradient
(HIP-event timing on the launch stream).  usage: python tools/bench_ph.py [size=128]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops

dev = torch.device("cuda")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128


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


for kind, cin, cout, sp in [("up", 64, 64, S // 2), ("up", 128, 128, S // 4), ("up", 256, 256, S // 8), ("s2", 32, 32, S), ("s2", 64, 64, S // 2),
                            ("s2", 128, 128, S // 4)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (cin * 27) ** 0.5
    plan = ops.UpConvPlan(1, (sp,) * 3, cin, cout) if kind == "up" else ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (2,) * 3, (1,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    dw = torch.zeros_like(w)
    flops = 2.0 * y.numel() * cin * 27  # the reference's count: 27 taps per output voxel
    t_p, t_f, t_d, t_w = timeit(lambda: plan.pack(w)), timeit(lambda: plan.fwd(x)), timeit(lambda: plan.dgrad(y)), timeit(lambda: plan.wgrad(x, y, dw))
    print(f"{kind} {cin:4d}->{cout:4d} in {sp:3d}^3: pack {t_p*1e6:7.1f} us | fwd {t_f*1e6:8.1f} us {flops/t_f/1e12:7.1f} TF | dgrad {t_d*1e6:8.1f} us "
          f"{flops/t_d/1e12:7.1f} TF | wgrad {t_w*1e6:8.1f} us {flops/t_w/1e12:7.1f} TF", flush=True)
