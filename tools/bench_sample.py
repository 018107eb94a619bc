"""Reverse-diffusion throughput of the C4 U-Net (128^3, batch 1): ms per denoising step (UNet forward + fused update, hipGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medical_image_generation_amd.inferer import DDPMScheduler, DiffusionInferer
from medical_image_generation_amd.unet import DiffusionModelUNet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda")
torch.manual_seed(0)
net = DiffusionModelUNet(**bench.C4)
for p in net.parameters():
    if float(p.detach().abs().max()) == 0:
        torch.nn.init.normal_(p, std=0.02)
net = net.to(dev).eval()
sch = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
sch.set_timesteps(nsteps)
inf = DiffusionInferer(sch)
x = torch.randn((1, 1, size, size, size), device=dev)
for graph in (True,):
    img = inf.sample(x, net, sch, verbose=False, use_graph=graph)   # includes plan creation + capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = inf.sample(x, net, sch, verbose=False, use_graph=graph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nsteps
    assert torch.isfinite(img).all()
    print({"config": f"C4 sampling {size}^3", "hipgraph": graph, "ms_per_denoising_step": dt * 1e3, "voxels_per_s": size ** 3 / dt,
           "s_per_1000_steps": dt * 1e3}, flush=True)
