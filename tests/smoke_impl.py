"""Body of __graft_entry__.smoke(): placeholder until the network path lands -- one fused conv vs torch fp32."""
import math

import torch
import torch.nn.functional as F


def run():
    from medical_image_generation_amd import hipops as ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(1, 32, 8, 8, 8).bfloat16().float()
    w = (torch.randn(32, 32, 3, 3, 3) / math.sqrt(27 * 32)).bfloat16().float()
    plan = ops.ConvPlan(1, (8, 8, 8), 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan.pack(w.to(dev))
    y = plan.fwd(x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16))
    ref = F.conv3d(x, w, padding=1)
    err = float((y.float().cpu().permute(0, 4, 1, 2, 3) - ref).abs().max())
    assert err <= 1e-2 * float(ref.abs().max()), err
