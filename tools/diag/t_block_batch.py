"""Probe: a stand-alone ResnetBlock (32 channels, 64^3) on batch 1 vs a batch of two identical samples."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from medical_image_generation_amd.blocks import ResnetBlock
torch.manual_seed(0)
for cin, cout in ((32, 32), (64, 32)):
    blk = ResnetBlock(3, cin, 128, cout)
    for p in blk.parameters():
        if float(p.detach().abs().max()) == 0: torch.nn.init.normal_(p, std=0.05)
    blk = blk.cuda()
    d = 64
    x1 = torch.randn(1, cin, d, d, d, device="cuda"); e1 = torch.randn(1, 128, device="cuda")
    with torch.no_grad():
        y1 = blk(x1, e1); y2 = blk(x1.repeat(2, 1, 1, 1, 1), e1.repeat(2, 1))
    print(cin, cout, "samples equal", bool(torch.equal(y2[0], y2[1])), "batch1 == batch2", bool(torch.equal(y2[0:1], y1)),
          "rel", float((y2[0:1] - y1).norm() / y1.norm()), "max abs", float((y2[0:1] - y1).abs().max()), flush=True)
