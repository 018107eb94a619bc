"""Single-launch latency of the 16^3-level weight gradient (own event pair per launch) -- run under MI_WGRAD_NSPLIT=1/2/4 and under
rocprofv3 --kernel-trace --stats for the main / reduce kernel durations."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops

dev = torch.device("cuda")
for cin, cout, sp in ((256, 256, 16), (512, 256, 16), (128, 128, 32), (64, 64, 64), (32, 32, 128)):
    x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
    y = torch.randn((1, sp, sp, sp, cout), device=dev).to(torch.bfloat16)
    plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (3,) * 3, (1,) * 3, (1,) * 3)
    dw = torch.zeros((cout, cin, 3, 3, 3), device=dev)
    ts = []
    st = torch.cuda.current_stream()
    for _ in range(32):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        plan.wgrad(x, y, dw)
        e1.record(st)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"nsplit={os.environ.get('MI_WGRAD_NSPLIT', 'auto')} {cin}->{cout}@{sp}: {statistics.median(ts[4:]):.1f} us", flush=True)
