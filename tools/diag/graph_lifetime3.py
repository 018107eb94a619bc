import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cases, synth
from medical_image_generation_amd.unet import DiffusionModelUNet
from medical_image_generation_amd.trainer import DDPMTrainer
S = cases.SEED
c = cases.UNET_CASES["unet3d"]
def mk():
    net = DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(synth.state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, S))
    return net.cuda()
x0 = synth.ellipsoid_volume(S, "x0", c["shape"]).cuda()
t = torch.tensor(c["timesteps"]).cuda()
noise = [synth.tensor(S, f"noise{k}", c["shape"]).cuda() for k in range(4)]
tg, te = DDPMTrainer(mk(), lr=1e-3), DDPMTrainer(mk(), lr=1e-3)
tg.capture(x0, noise[0], t)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
def show(k):
    a, b = float(tg.step_graph(x0, noise[k], t)), float(te.step(x0, noise[k], t))
    n = te.arena.n_trainable
    print(k, f"loss {a:.7f} {b:.7f}", "grad", f"{rel(tg.arena.grad[:n], te.arena.grad[:n]):.2e}", "sumsq", float(tg.sumsq), float(te.sumsq), "step", float(tg.step_count), float(te.step_count),
          "m", f"{rel(tg.exp_avg, te.exp_avg):.2e}", "v", f"{rel(tg.exp_avg_sq, te.exp_avg_sq):.2e}", "p", f"{rel(tg.arena.data[:n], te.arena.data[:n]):.2e}", flush=True)
show(0)
z = torch.zeros(1024, device="cuda")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    z += 1
torch.cuda.synchronize()
for k in (1, 2): show(k)
