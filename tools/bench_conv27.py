"""Quick A/B of the k3 s1 kernels on a few shapes with the in-step epilogue variants (bias rows, residual).
usage: python tools/bench_conv27.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")

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
    return e0.elapsed_time(e1) / iters * 1e3

shapes = [(32, 32, 128), (64, 64, 128), (96, 32, 128), (128, 128, 64), (256, 256, 16)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for cin, cout, sp in shapes:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (cin * 27) ** 0.5
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    res = torch.randn_like(y)
    av = torch.randn((1, cout), device=dev)
    fl = 2.0 * y.numel() * cin * 27
    t0 = timeit(lambda: plan.fwd(x))
    t1 = timeit(lambda: plan.fwd(x, addvec=av))
    t2 = timeit(lambda: plan.fwd(x, addvec=av, res=res))
    t3 = timeit(lambda: plan.dgrad(y))
    print(f"{cin:4d}->{cout:4d} @{sp:3d}: fwd {t0:7.1f} us {fl/t0/1e6:6.0f} TF | +addvec {t1:7.1f} | +addvec+res {t2:7.1f} | dgrad {t3:7.1f} us {fl/t3/1e6:6.0f} TF", flush=True)
