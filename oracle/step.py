"""Oracle train-step glue: DDPM schedule closed forms, losses, one optimizer step.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference takes these from `monai-generative` (`generative.networks.schedulers.
DDPMScheduler`, T-LDM:74,160,165; T-DDPM:380), which is absent from /root/reference and
from this image (unpinned dependency, pyproject.toml:20-33).  The formulas below are the
published DDPM / LDM closed forms; nothing under /root/reference pins them, so the
schedule itself is "parity unpinned" (DESIGN.md).  The step order follows T-LDM:145-180 /
T-DDPM:183-200; optimizer and clipping are torch's own (`torch.optim.Adam[W]`,
`clip_grad_norm_`), exactly what the reference calls.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class DDPMSchedule:
    """betas / alphas_cumprod of DDPMScheduler(num_train_timesteps, schedule, beta_start, beta_end)."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "scaled_linear_beta",
                 beta_start: float = 0.0015, beta_end: float = 0.0205, prediction_type: str = "epsilon"):
        if schedule == "scaled_linear_beta":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif schedule == "linear_beta":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"unknown schedule {schedule}")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    def _coef(self, timesteps, ndim):
        a = self.alphas_cumprod[timesteps]
        shape = (-1,) + (1,) * (ndim - 1)
        return (a ** 0.5).reshape(shape), ((1 - a) ** 0.5).reshape(shape)

    def add_noise(self, original_samples, noise, timesteps):
        sa, so = self._coef(timesteps, original_samples.ndim)
        return sa * original_samples + so * noise

    def step_coefficients(self):
        """Per-timestep constants of DDPMScheduler.step (third-party `generative.networks.schedulers.ddpm`, not under /root/reference:
        restated from Ho et al. 2020 eq. 6-7 as upstream implements it; PARITY UNPINNED -- the reference holds no vectors for it).
        Returns [T, 5]: 1/sqrt(acp_t), sqrt(1-acp_t), c_x0 = sqrt(acp_{t-1}) beta_t / (1-acp_t), c_xt = sqrt(alpha_t) (1-acp_{t-1}) / (1-acp_t),
        sigma_t = sqrt(clamp((1-acp_{t-1})/(1-acp_t) beta_t, 1e-20)) for t > 0 and 0 at t = 0 ("fixed_small")."""
        acp = self.alphas_cumprod.double()
        betas = self.betas.double()
        prev = torch.cat([torch.ones(1, dtype=torch.float64), acp[:-1]])
        c_x0 = prev.sqrt() * betas / (1 - acp)
        c_xt = (1 - betas).sqrt() * (1 - prev) / (1 - acp)
        var = ((1 - prev) / (1 - acp) * betas).clamp(min=1e-20)
        sigma = var.sqrt()
        sigma[0] = 0.0
        return torch.stack([1 / acp.sqrt(), (1 - acp).sqrt(), c_x0, c_xt, sigma], dim=1).float()

    def step(self, model_output, timestep: int, sample, noise, clip_sample: bool = True):
        """x_{t-1} and the predicted x_0 (epsilon prediction), noise = the z of this step."""
        k = self.step_coefficients()[timestep]
        if self.prediction_type == "v_prediction":
            x0 = sample / k[0] - k[1] * model_output
        else:
            x0 = (sample - k[1] * model_output) * k[0]
        if clip_sample:
            x0 = x0.clamp(-1, 1)
        prev = k[2] * x0 + k[3] * sample
        if timestep > 0:
            prev = prev + k[4] * noise
        return prev, x0

    def get_velocity(self, sample, noise, timesteps):
        sa, so = self._coef(timesteps, sample.ndim)
        return sa * noise - so * sample


def kl_loss(z_mu, z_sigma):
    """AutoEncoder.get_kl_loss, T-AE:68-72."""
    kl = 0.5 * (z_mu.pow(2) + z_sigma.pow(2) - torch.log(z_sigma.pow(2)) - 1)
    kl = torch.sum(kl, dim=list(range(1, z_mu.ndim)))
    return torch.sum(kl) / kl.shape[0]


def ddpm_loss(model, schedule: DDPMSchedule, x0, noise, timesteps, condition=None):
    """q-sample -> model -> MSE (T-LDM:159-169 / T-DDPM:185-192).

    condition: the `condition=..., mode="concat"` arguments of the inferer call at T-DDPM:191 (third-party
    `generative.inferers.DiffusionInferer.__call__`, source absent: restated from upstream -- PARITY UNPINNED):
    the un-noised condition is concatenated behind the NOISED image on the channel axis, the model predicts
    the image's channels and the target is the image's noise (BASELINE configs[4])."""
    noisy = schedule.add_noise(x0, noise, timesteps)
    if condition is not None:
        noisy = torch.cat([noisy, condition], dim=1)
    pred = model(noisy, timesteps)
    target = schedule.get_velocity(x0, noise, timesteps) if schedule.prediction_type == "v_prediction" else noise
    return F.mse_loss(pred.float(), target.float()), pred


def ddpm_train_step(model, optimizer, schedule, x0, noise, timesteps, max_norm: float | None = 1.0, condition=None):
    """backward -> clip_grad_norm_ -> step -> zero_grad (T-LDM:171-180)."""
    loss, pred = ddpm_loss(model, schedule, x0, noise, timesteps, condition)
    loss.backward()
    if max_norm:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss.detach(), pred.detach()


def ae_loss(model, images, eps, kl_weight: float):
    """Generator-side AE loss without the third-party perceptual/adversarial terms
    (T-AE:411-414): L1 reconstruction + kl_weight * KL."""
    recon, z_mu, z_sigma = model(images, eps)
    rec = F.l1_loss(recon.float(), images.float())
    reg = kl_loss(z_mu, z_sigma) * kl_weight
    return rec + reg, recon, z_mu, z_sigma
