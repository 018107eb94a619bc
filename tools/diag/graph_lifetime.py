"""Diagnostic: captured trainer vs eager trainer with a sampling detour at another shape in between."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cases, nets, synth
from medical_image_generation_amd.unet import DiffusionModelUNet
from medical_image_generation_amd.trainer import DDPMTrainer
from medical_image_generation_amd.inferer import DDPMScheduler, DiffusionInferer
S = cases.SEED
c = cases.UNET_CASES["unet3d"]
def mk():
    net = DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, S)
    net.load_state_dict(sd)
    return net.cuda()
x0 = synth.ellipsoid_volume(S, "x0", c["shape"]).cuda()
t = torch.tensor(c["timesteps"]).cuda()
noise = [synth.tensor(S, f"noise{k}", c["shape"]).cuda() for k in range(4)]
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
trs = {"graph+detour": DDPMTrainer(mk(), lr=1e-3), "eager": DDPMTrainer(mk(), lr=1e-3), "graph": DDPMTrainer(mk(), lr=1e-3), "eager+detour": DDPMTrainer(mk(), lr=1e-3)}
for k in ("graph+detour", "graph"):
    trs[k].capture(x0, noise[0], t)
def step(name, k):
    tr = trs[name]
    return float(tr.step_graph(x0, noise[k], t)) if name.startswith("graph") else float(tr.step(x0, noise[k], t))
def show(k):
    ls = {n: step(n, k) for n in trs}
    ref = trs["eager"].arena.data[:trs["eager"].arena.n_trainable]
    d = {n: float((trs[n].arena.data[:ref.numel()] - ref).norm() / ref.norm()) for n in trs}
    print(k, {n: f"{v:.7f}" for n, v in ls.items()}, {n: f"{v:.2e}" for n, v in d.items()}, flush=True)
show(0)
sch = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
sch.set_timesteps(4)
inf = DiffusionInferer(sch)
big = (5,) + tuple(c["shape"][1:3]) + (24, 24)
for n in ("graph+detour", "eager+detour"):
    inf.sample(torch.randn(big, device="cuda"), trs[n].model, sch, verbose=False, use_graph=(mode == "graph"))
gc.collect(); torch.cuda.synchronize()
for k in (1, 2, 3):
    show(k)
