"""Warm vs in-step-like timing of the coarse-level convs: every launch is bracketed by its own pair of HIP events.
  warm : the same launch repeated (weights, operands and outputs in L2 / Infinity Cache)
  cold : a 1 GiB fill runs before every launch (nothing on chip), then the ACTIVATION operands are re-touched by a small copy -- what a
         layer meets in the step: operands just produced, packed weights / gradient slabs last touched ~20 ms ago
usage: python tools/bench_cold.py [reps=20]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops

dev = torch.device("cuda")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [(256, 256, 16), (512, 256, 16), (128, 128, 32), (256, 128, 32), (64, 64, 64), (32, 32, 128)]
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def per_launch(fn, prep=None):
    ts = []
    st = torch.cuda.current_stream()
    for _ in range(REPS + 2):
        if prep is not None:
            prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts[2:])


for cin, cout, sp in SHAPES:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (cin * 27) ** 0.5
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    dw = torch.zeros_like(w)
    x2, y2 = torch.empty_like(x), torch.empty_like(y)

    def cold_x():
        flush.fill_(1)
        x2.copy_(x)
        x.copy_(x2)

    def cold_y():
        flush.fill_(1)
        y2.copy_(y)
        y.copy_(y2)

    def cold_xy():
        flush.fill_(1)
        x2.copy_(x)
        x.copy_(x2)
        y2.copy_(y)
        y.copy_(y2)

    flops = 2.0 * y.numel() * cin * 27
    r = {}
    for name, fn, prep in (("fwd", lambda: plan.fwd(x), cold_x), ("dgrad", lambda: plan.dgrad(y), cold_y),
                           ("wgrad", lambda: plan.wgrad(x, y, dw), cold_xy)):
        r[name] = (per_launch(fn), per_launch(fn, prep))
    print(f"{cin:4d}->{cout:4d} @{sp:3d}^3: " + " | ".join(f"{k} warm {a:6.1f} cold {b:6.1f} us ({flops / b / 1e6:5.0f} TF)" for k, (a, b) in r.items()),
          flush=True)
