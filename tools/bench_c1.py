"""Timing of the single-channel input / output convs at 128^3 (conv_c1.hip vs MI_CONV_C1=0: the implicit-GEMM kernels)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout in ((1, 32), (32, 1)):
    plan = ops.ConvPlan(1, (S, S, S), cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
    plan.pack(w)
    x = torch.randn(1, S, S, S, cin, device=dev).bfloat16()
    dy = torch.randn(1, S, S, S, cout, device=dev).bfloat16()
    b = torch.zeros(cout, device=dev); dw = torch.zeros_like(w); cb = torch.zeros(cout, device=dev)
    print(f"{cin}->{cout} @{S}: fwd {t(lambda: plan.fwd(x, addvec=b)):.1f} us  dgrad {t(lambda: plan.dgrad(dy)):.1f} us  "
          f"wgrad {t(lambda: plan.wgrad(x, dy, dw, colsum=cb)):.1f} us", flush=True)
