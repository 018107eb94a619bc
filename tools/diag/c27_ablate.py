"""One arm of the k_conv27 ablation (tools/diag/c27_ablate.sh): forward 32->32 and 64->64 at 128^3, random data.
Prints wall time per launch; with MI_C27_DBG=64 the library prints the main-loop shader cycles of workgroup 0 on stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for cin, cout, sp in [(32, 32, 128), (64, 64, 128)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / (cin * 27) ** 0.5
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    plan.pack(w)
    for _ in range(2):
        y = plan.fwd(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.fwd(x)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * y.numel() * cin * 27
    nm = fl / 32768 / 1024  # MFMAs per compute wave
    print(f"{os.environ.get('MI_LIB_PATH','base').split('_')[-1]:>8s} dbg={os.environ.get('MI_C27_DBG','0'):>3s} {cin}->{cout}: {t:7.1f} us  {fl/t/1e6:6.0f} TF  ({nm:.0f} MFMAs per wave)", flush=True)
