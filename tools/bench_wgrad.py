import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
for cin, cout, sp in [(32, 32, 128), (64, 64, 128), (96, 32, 128), (64, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16), (192, 64, 64)]:
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) / 30
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    plan.pack(w)
    y = plan.fwd(x)
    dw = torch.zeros_like(w)
    for _ in range(2):
        plan.wgrad(x, y, dw)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        plan.wgrad(x, y, dw)
    e1.record(st); e1.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"roll={os.environ.get('MI_WGRAD_ROLL','1')} dbg={os.environ.get('MI_WGRAD_DBG','0')} {cin}->{cout}@{sp}: {t*1e3:.1f} us  {2.0*sp**3*cin*cout*27/t/1e9:.0f} TF", flush=True)
