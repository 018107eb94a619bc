"""Counts and GPU time of the library's entry points in ONE eager C4 train step at 128^3 (which copies / adds are left?)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from medical_image_generation_amd import _lib
from medical_image_generation_amd.trainer import DDPMTrainer
from medical_image_generation_amd.unet import DiffusionModelUNet
dev = torch.device("cuda")
torch.manual_seed(0)
net = DiffusionModelUNet(**bench.C4).to(dev)
tr = DDPMTrainer(net, lr=2e-5)
x0 = bench.synthetic_volume((1, 1, 128, 128, 128), 1, dev)
noise = torch.randn_like(x0); t = torch.randint(0, 1000, (1,), device=dev)
tr.step(x0, noise, t)
with _lib.profile_calls() as prof:
    tr.step(x0, noise, t)
torch.cuda.synchronize()
cnt, ms = collections.Counter(), collections.Counter()
for name, e0, e1 in prof.records:
    cnt[name] += 1; ms[name] += e0.elapsed_time(e1)
for k, v in sorted(ms.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {cnt[k]:4d} calls {v:8.3f} ms")
