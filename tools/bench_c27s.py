import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
def timeit(fn, iters=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for cin, cout, sp in [(32, 32, 128), (64, 64, 128), (64, 32, 128)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (cin * 27) ** 0.5
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3); plan.pack(w)
    y = plan.fwd(x); res = torch.randn_like(y); av = torch.randn((1, cout), device=dev)
    dw = torch.zeros_like(w); cb = torch.zeros(cout, device=dev)
    for rep in range(2):
        r = [timeit(lambda: plan.fwd(x, addvec=av)), timeit(lambda: plan.fwd(x, addvec=av, want_sums=True)),
             timeit(lambda: plan.fwd(x, addvec=av, res=res)), timeit(lambda: plan.fwd(x, addvec=av, res=res, want_sums=True)),
             timeit(lambda: plan.dgrad(y)), timeit(lambda: plan.wgrad(x, y, dw, colsum=cb))]
        print(f"{cin}->{cout}@{sp} rep{rep}: fwd {r[0]:.1f} | +sums {r[1]:.1f} | +res {r[2]:.1f} | +res+sums {r[3]:.1f} | dgrad {r[4]:.1f} | wgrad {r[5]:.1f}", flush=True)
