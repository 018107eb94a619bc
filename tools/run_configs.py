"""Run the other BASELINE.json configs through the fused HIP trainer and report ms/step (sanity + scale check).
usage: python tools/run_configs.py c2|c3a|c3a_gan|c3b|c3b_ldm|c5|c5_ckpt [steps]
  c3a_gan: the autoencoder step AFTER the warm-up epochs -- generator step with the adversarial term through the planner's PatchDiscriminator
           (64 base channels, 3 layers) + the discriminator step (AEGANTrainer), both networks on the HIP path
  c3b_ldm: the latent-diffusion step as train_ldm.py runs it -- no-grad AutoencoderKL.encode_stage_2_inputs of the 4 x 128^3 images
           inside the step (LDMTrainer), then the C3b UNet step on the scaled latents
  c5_ckpt: C5 with per-block activation checkpointing (BASELINE configs[4])"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_generation_amd import engine as E
from medical_image_generation_amd.trainer import DDPMTrainer
from medical_image_generation_amd.unet import DiffusionModelUNet

which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def run_c3a(gan=False):
    """AutoencoderKL exactly as CFG:821-862 emits it for a 128^3 single-channel dataset (BASELINE configs[2], AE half):
    fused generator step (L1 + kl_weight*KL, clip 1, Adam; T-AE:411-434) replayed from a hipGraph."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import AEGANTrainer, AETrainer
    down = [[[1] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3]]
    kw = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, num_res_blocks=2, num_channels=[32, 64, 128],
              attention_levels=[False] * 3, norm_num_groups=16, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False,
              downsample_parameters=down, upsample_parameters=list(reversed(down))[:-1])
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = AutoencoderKL(**kw).to(dev)
    if gan:
        from medical_image_generation_amd.discriminator import PatchDiscriminator
        disc = PatchDiscriminator(spatial_dims=3, in_channels=1, out_channels=1, num_channels=64, num_layers_d=3).to(dev)  # CFG:966-967
        tr = AEGANTrainer(net, disc, adv_weight=0.01, lr=5e-5, d_lr=5e-5, kl_weight=1e-7)
    else:
        tr = AETrainer(net, lr=5e-5, kl_weight=1e-7)
    x = torch.rand((2, 1, 128, 128, 128), device=dev)
    eps = torch.randn((2, 8, 32, 32, 32), device=dev)
    tr.capture(x, eps)
    for _ in range(2):
        tr.step_graph()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step_graph()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert math.isfinite(float(loss)), "non-finite loss: not a measurement"
    print(json.dumps({"config": "c3a_gan" if gan else "c3a", "ms_per_step": dt * 1e3, "voxels_per_s": 2 * 128 ** 3 / dt, "loss": float(loss),
                      "params": sum(p.numel() for p in net.parameters()), "model_flops_per_step_survey": 1.497e13,
                      "mfma_frac": 1.497e13 / dt / 2.5e15, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)


if which in ("c3a", "c3a_gan"):
    run_c3a(gan=which == "c3a_gan")
    sys.exit(0)
iso = lambda n: [[1] * 3] + [[2] * 3] * (n - 1)
if which == "c2":   # 3D DDPM 96^3, batch 2 (BASELINE configs[1])
    kw = dict(spatial_dims=3, in_channels=1, out_channels=1, num_res_blocks=2, num_channels=(32, 64, 128, 256),
              attention_levels=(False, False, False, True), num_head_channels=(0, 0, 0, 64), norm_num_groups=32,
              strides=iso(4), kernel_sizes=[[3] * 3] * 4, paddings=[[1] * 3] * 4)
    shape, vox_per_sample = (2, 1, 96, 96, 96), 96 ** 3
elif which in ("c3b", "c5", "c3b_ldm", "c5_ckpt"):  # latent UNet the reference's planner emits (CFG:876-902); C5: 40^3 latents (160^3 patch), +1 label channel
    ldm, ckpt = which == "c3b_ldm", which == "c5_ckpt"
    which = {"c3b_ldm": "c3b", "c5_ckpt": "c5"}.get(which, which)
    cin = 8 if which == "c3b" else 9
    kw = dict(spatial_dims=3, in_channels=cin, out_channels=8, num_res_blocks=2, num_channels=[256, 512, 768],
              attention_levels=[False, True, True], num_head_channels=[0, 512, 768], norm_num_groups=32,
              strides=iso(3), kernel_sizes=[[3] * 3] * 3, paddings=[[1] * 3] * 3)
    shape = (4, 8, 32, 32, 32) if which == "c3b" else (1, 8, 40, 40, 40)  # (C5: + 1 un-noised label channel, `condition`)
    vox_per_sample = 128 ** 3 if which == "c3b" else 160 ** 3
dev = torch.device("cuda")
torch.manual_seed(0)
net = DiffusionModelUNet(**kw)
for p in net.parameters():
    if float(p.detach().abs().max()) == 0:
        torch.nn.init.normal_(p, std=0.02)
net = net.to(dev)
ldm, ckpt = globals().get("ldm", False), globals().get("ckpt", False)
net.use_checkpointing = ckpt
noise = torch.randn(shape, device=dev)
t = torch.randint(0, 1000, (shape[0],), device=dev)
if ldm:
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import LDMTrainer
    down = [[[1] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3]]
    ae = AutoencoderKL(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, num_res_blocks=2, num_channels=[32, 64, 128],
                       attention_levels=[False] * 3, norm_num_groups=16, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False,
                       downsample_parameters=down, upsample_parameters=list(reversed(down))[:-1]).to(dev)
    tr = LDMTrainer(net, ae, lr=2e-5)
    images, eps = torch.rand((4, 1, 128, 128, 128), device=dev), torch.randn(shape, device=dev)
    tr.estimate_scale_factor(images, eps)
    tr.capture(images, eps, noise, t)
else:
    tr = DDPMTrainer(net, lr=2e-5)
    x0 = torch.rand(shape, device=dev)
    if which == "c5":  # BASELINE configs[4]: label-channel conditioning = a binary mask concatenated un-noised behind the 8 noised latents
        cond = (torch.rand((shape[0], 1) + shape[2:], device=dev) > 0.5).float()
        tr.capture(x0, noise, t, None, None, cond)
    else:
        tr.capture(x0, noise, t)
for _ in range(2):
    tr.step_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step_graph()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
c = E.Ctx(tr.arena, net._plans, grad_enabled=True)
dims = (shape[0],) + tuple(shape[2:]) + (kw["in_channels"],)
net._run(c, torch.zeros(dims, dtype=torch.bfloat16, device=dev), t, need_dx=False)
c.tape.fns.clear()
fl = c.flops_fwd + c.flops_bwd
assert math.isfinite(float(loss)), "non-finite loss: not a measurement"
which = which + ("_ldm" if ldm else "") + ("_ckpt" if ckpt else "")
print(json.dumps({"config": which, "ms_per_step": dt * 1e3, "voxels_per_s": shape[0] * vox_per_sample / dt, "loss": float(loss),
                  "params": sum(p.numel() for p in net.parameters()), "model_flops_per_step": fl, "mfma_frac": fl / dt / 2.5e15,
                  "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
