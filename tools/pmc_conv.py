"""Tiny driver for PMC runs: a few launches of one conv kernel.  usage: pmc_conv.py fwd|wgrad cin cout size [kernel=3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import hipops as ops
mode, cin, cout, sp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda")
x = torch.randn((1, sp, sp, sp, cin), device=dev).to(torch.bfloat16)
w = torch.randn((cout, cin, k, k, k), device=dev) / (cin * k ** 3) ** 0.5
plan = ops.ConvPlan(1, (sp,) * 3, cin, cout, (k,) * 3, (1,) * 3, (k // 2,) * 3)
plan.pack(w)
y = plan.fwd(x)
dw = torch.zeros_like(w)
for _ in range(3):
    if mode == "fwd":
        plan.fwd(x)
    else:
        plan.wgrad(x, y, dw)
torch.cuda.synchronize()
