"""time the single-channel weight gradients: python tools/bench_c1w.py [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
sp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for cin, cout in [(1, 32), (32, 1)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.1
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3); plan.pack(w)
    y = plan.fwd(x)
    dw = torch.zeros_like(w); cb = torch.zeros(cout, device=dev)
    print(f"{cin}->{cout}@{sp}: fwd {timeit(lambda: plan.fwd(x)):.1f} us | dgrad {timeit(lambda: plan.dgrad(y)):.1f} | wgrad {timeit(lambda: plan.wgrad(x, y, dw, colsum=cb)):.1f}", flush=True)
