"""Reverse-diffusion sampling (medical_image_generation_amd/inferer.py) against the CPU restatement: the fused update kernel against
DDPMSchedule.step (oracle/step.py; third-party closed form, parity unpinned -- see its docstring), and the whole sample loop
(UNet forward + update, hipGraph-replayed) against the same loop on the oracle's UNet with the per-step noise pinned."""
import pytest
import torch

from oracle import cases, nets, step, synth

pytestmark = pytest.mark.gpu
S = cases.SEED


def test_ddpm_step_kernel_matches_closed_form():
    from medical_image_generation_amd import hipops as ops
    from medical_image_generation_amd._lib import call, ptr
    from medical_image_generation_amd.inferer import DDPMScheduler
    sch = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
    ref = step.DDPMSchedule()
    assert torch.allclose(sch._coef, ref.step_coefficients(), rtol=1e-6, atol=1e-7)
    shape = (2, 3, 4, 6, 5)
    for t in (999, 500, 1, 0):
        x, z = synth.tensor(S, f"x{t}", shape) * 1.5, synth.tensor(S, f"z{t}", shape)
        eps = synth.tensor(S, f"e{t}", shape).bfloat16().float()
        want, _ = ref.step(eps, t, x, z, clip_sample=True)
        xd, zd = x.cuda(), z.cuda()
        eps_cl = ops.to_channels_last(eps.cuda())
        x_cl = torch.empty_like(eps_cl)
        td = torch.tensor([t], device="cuda")
        call("mi_ddpm_step", ptr(xd), ptr(eps_cl), ptr(zd), ptr(sch.coefficients("cuda")), ptr(td), ptr(x_cl), shape[0], shape[1], 4 * 6 * 5, 1)
        assert float((xd.cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
        assert float((ops.to_channels_first(x_cl).cpu() - want).abs().max()) <= 1e-2 * float(want.abs().max())
    # upstream's tensor-level step() signature
    prev, x0 = sch.step(eps, 0, x)
    w_prev, w_x0 = ref.step(eps, 0, x, None)
    assert torch.allclose(prev, w_prev, atol=1e-5) and torch.allclose(x0, w_x0, atol=1e-5)
    sch.set_timesteps(10)
    assert sch.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    with pytest.raises(ValueError):
        sch.set_timesteps(2000)


@pytest.mark.parametrize("graph", [False, True])
def test_sample_loop_matches_oracle(graph):
    from medical_image_generation_amd.inferer import DDPMScheduler, DiffusionInferer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES["unet3d"]
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(sd)
    net = net.cuda().eval()
    sch = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
    sch.set_timesteps(5)
    steps = sch.timesteps.tolist()
    shape = c["shape"]
    x0 = synth.tensor(S, "sample_noise", shape)
    zs = [synth.tensor(S, f"sample_z{i}", shape) for i in range(len(steps))]
    oracle = step.DDPMSchedule()
    x = x0.clone()
    with torch.no_grad():
        for i, t in enumerate(steps):
            eps = ref(x, torch.full((shape[0],), t, dtype=torch.int64))
            x, _ = oracle.step(eps, t, x, zs[i], clip_sample=True)
    got = DiffusionInferer(sch).sample(x0.cuda(), net, sch, verbose=False, noises=[z.cuda() for z in zs], use_graph=graph)
    err = float((got.cpu() - x).norm() / x.norm())
    print(f"\n[sample loop graph={graph}] rel-L2 after {len(steps)} steps: {err:.3e}")
    assert torch.isfinite(got).all() and err <= 3e-2  # bf16 UNet forward, 5 steps (same budget as the forward parity tests)
