import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for cin, cout, sp in [(32, 32, 128), (64, 64, 64), (256, 256, 16), (128, 128, 32)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev)
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    dw = torch.zeros_like(w)
    cb = torch.zeros(cout, device=dev)
    a = timeit(lambda: plan.wgrad(x, y, dw)); b = timeit(lambda: plan.wgrad(x, y, dw, colsum=cb)); c = timeit(lambda: plan.wgrad(x, y, dw)); d = timeit(lambda: plan.wgrad(x, y, dw, colsum=cb))
    print(f"{cin}->{cout}@{sp}: no colsum {a:.1f} / {c:.1f} us | bias colsum {b:.1f} / {d:.1f}", flush=True)
