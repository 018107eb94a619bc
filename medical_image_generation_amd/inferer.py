"""Reverse-diffusion sampling on the HIP path: the loop behind `inferer.sample(...)` in LDM.sample_images (train_ldm.py:332-366) and
DDPM.sample_image (train_ddpm.py:238-246).

The reference takes `DDPMScheduler`, `DiffusionInferer` and `LatentDiffusionInferer` from the third-party `generative` package
(train_ldm.py:74, 102, 112), which is not under /root/reference; the classes here mirror the part of their interface those call
sites use (same names, argument meaning, return values).  The scheduler arithmetic is the closed form of Ho et al. 2020 as upstream
implements it, restated in oracle/step.py (`DDPMSchedule.step`): PARITY UNPINNED -- the reference holds no vectors for it; the
tests pin the kernel against that restatement.

One denoising step = UNet forward (no tape) + ONE fused update kernel (`mi_ddpm_step`: predicted x0, clip, posterior mean, noise;
it also writes the next step's channels-last bf16 model input), captured once as a hipGraph and replayed per timestep -- the
timestep is a device scalar the embedding and update kernels read, so all 1000 steps share one graph.
"""
from __future__ import annotations

import torch

from . import engine as E
from . import hipops as ops
from ._lib import call, ptr

F32 = torch.float32


class DDPMScheduler:
    """`generative.networks.schedulers.DDPMScheduler` as train_ldm.py:74 constructs it (variance_type "fixed_small")."""

    def __init__(self, num_train_timesteps=1000, schedule="linear_beta", variance_type="fixed_small", clip_sample=True,
                 prediction_type="epsilon", beta_start=1e-4, beta_end=2e-2):
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' (the upstream default the reference uses)")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"unknown prediction_type {prediction_type}")
        if schedule == "scaled_linear_beta":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=F32) ** 2
        elif schedule == "linear_beta":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=F32)
        else:
            raise ValueError(f"unknown schedule {schedule}")
        self.num_train_timesteps, self.prediction_type, self.clip_sample = num_train_timesteps, prediction_type, clip_sample
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        acp, b = self.alphas_cumprod.double(), betas.double()
        prev = torch.cat([torch.ones(1, dtype=torch.float64), acp[:-1]])
        sigma = ((1 - prev) / (1 - acp) * b).clamp(min=1e-20).sqrt()
        sigma[0] = 0.0  # no noise is added at t = 0
        self._coef = torch.stack([1 / acp.sqrt(), (1 - acp).sqrt(), prev.sqrt() * b / (1 - acp), (1 - b).sqrt() * (1 - prev) / (1 - acp), sigma],
                                 dim=1).float().contiguous()
        self._coef_dev = None
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.num_train_timesteps`: "
                             f"{self.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).to(torch.int64)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def coefficients(self, device):
        if self._coef_dev is None or self._coef_dev.device != torch.device(device):
            self._coef_dev = self._coef.to(device)
        return self._coef_dev

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod.to(original_samples.device)[timesteps]
        shape = (-1,) + (1,) * (original_samples.ndim - 1)
        return (a ** 0.5).reshape(shape) * original_samples + ((1 - a) ** 0.5).reshape(shape) * noise

    def step(self, model_output, timestep, sample, generator=None):
        """(pred_prev_sample, pred_original_sample) for fp32 NC[D]HW tensors -- the upstream signature; the fused loop of the inferers
        below does not go through this (tensor-level) method."""
        t = int(timestep)
        k = self._coef[t].to(sample.device)
        if self.prediction_type == "v_prediction":
            x0 = sample / k[0] - k[1] * model_output
        else:
            x0 = (sample - k[1] * model_output) * k[0]
        if self.clip_sample:
            x0 = x0.clamp(-1, 1)
        prev = k[2] * x0 + k[3] * sample
        if t > 0:
            prev = prev + k[4] * torch.randn(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        return prev, x0


class _Loop:
    """One captured denoising step, replayed over scheduler.timesteps."""

    def __init__(self, model, scheduler, shape, device, cond_channels=0, context_shape=None):
        """cond_channels: channels of a mode="concat" conditioning tensor that sit, un-noised, behind the sample's channels in the model
        input; context_shape: (tokens, cross_attention_dim) of a mode="crossattn" conditioning."""
        self.m, self.sch = model, scheduler
        n, c = shape[0], shape[1]
        sp = tuple(shape[2:])
        self.v = 1
        for s in sp:
            self.v *= s
        dims = (1,) * (3 - len(sp)) + sp
        self.n, self.c = n, c
        self.x = torch.empty(shape, dtype=F32, device=device)                    # the sample, fp32 NC[D]HW
        self.cc = int(cond_channels)
        self.x_cl = torch.empty((n,) + dims + (c + self.cc,), dtype=torch.bfloat16, device=device)  # ... as the model reads it (+ condition)
        self.ctx = None if context_shape is None else torch.empty((n * context_shape[0], context_shape[1]), dtype=torch.bfloat16, device=device)
        self.z = torch.empty(shape, dtype=F32, device=device)
        self.t = torch.zeros(1, dtype=torch.int64, device=device)
        self.tn = torch.zeros(n, dtype=torch.int64, device=device)
        self.coef = scheduler.coefficients(device)
        self.arena = model.arena(device)
        self.graph = None
        self.keys = ()  # conv plans whose weights are packed: once per run(), not per step (the weights do not change while sampling)
        self._pinned = None  # objects whose device memory the captured graph points at (conv plans, GroupNorm workspace)

    def _step(self):
        ctx = E.Ctx(self.arena, self.m._plans, grad_enabled=False, prepacked=self.keys)
        eps = self.m._run(ctx, self.x_cl, self.tn, need_dx=False, context=self.ctx)
        call("mi_ddpm_step", ptr(self.x), ptr(eps), ptr(self.z), ptr(self.coef), ptr(self.t), ptr(self.x_cl), self.c + self.cc, self.n, self.c,
             self.v, int(self.sch.clip_sample) | (2 if self.sch.prediction_type == "v_prediction" else 0))

    def _load(self, x):
        """x (fp32 NC[D]HW) -> the sample channels of the model input (the condition channels behind them stay)."""
        self.x.copy_(x)
        self.x_cl[..., :self.c].copy_(ops.to_channels_last(self.x))

    def run(self, input_noise, noises=None, generator=None, use_graph=True, on_step=None, conditioning=None):
        if self.m._arena is not self.arena:  # the module moved (.to / .cuda): plans and graph point at the old buffers -> start over
            self.arena, self.graph, self.keys, self._pinned = self.m.arena(self.x.device), None, (), None
        if self.cc:  # mode="concat": model_input = cat([image, conditioning], dim=1) at every step; written once, the steps rewrite `image`
            self.x_cl[..., self.c:].copy_(ops.to_channels_last(conditioning))
        elif self.ctx is not None:  # mode="crossattn": the context token matrix is a constant of the loop
            self.ctx.copy_(ops.cast_bf16(conditioning.contiguous().reshape(-1, conditioning.shape[2])))
        self._load(input_noise)
        steps = [int(t) for t in self.sch.timesteps]
        keep = self.x.clone()
        if not self.m._plans or len(self.keys) != len(self.m._plans):
            self.keys = ()
            self.z.zero_()
            self._step()  # first pass at this shape: creates the conv plans, every conv packs its own weights
        self.keys = self.m.pack_all()  # current weights, one launch
        if use_graph and self.graph is None and len(steps) > 2:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.z.zero_()
                self._step()  # workspaces (hipMalloc) outside the capture
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._step()
            self._pinned = (dict(self.m._plans), dict(ops._ws_cache), self.arena)  # keep what the graph points at alive
        self._load(keep)
        for i, t in enumerate(steps):
            self.t.fill_(t)
            self.tn.fill_(t)
            if noises is not None:
                self.z.copy_(noises[i])
            elif t > 0:
                self.z.normal_(generator=generator)
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
            if on_step is not None:
                on_step(i, t, self.x)
        return self.x.clone()


class DiffusionInferer:
    """`generative.inferers.DiffusionInferer(scheduler)`: `sample(input_noise, diffusion_model, scheduler)` (train_ddpm.py:243)."""

    def __init__(self, scheduler):
        self.scheduler = scheduler
        self._loops = {}

    def _loop(self, model, scheduler, shape, device, cond_channels=0, context_shape=None):
        key = (id(model), id(scheduler), tuple(shape), cond_channels, context_shape)
        if key not in self._loops:
            self._loops[key] = _Loop(model, scheduler, tuple(shape), device, cond_channels, context_shape)
        return self._loops[key]

    @staticmethod
    def _check_mode(mode):
        if mode not in ("crossattn", "concat"):
            raise NotImplementedError(f"{mode} condition is not supported")

    def __call__(self, inputs, diffusion_model, noise, timesteps, condition=None, mode="crossattn"):
        """The training-side call of train_ddpm.py:191: `noise_pred = inferer(inputs=images, diffusion_model=model, noise=noise,
        timesteps=timesteps)` = diffusion_model(scheduler.add_noise(inputs, noise, timesteps), timesteps) (third-party
        `generative.inferers.DiffusionInferer.__call__`).  Differentiable through the model's autograd edge; the fused
        trainer.DDPMTrainer is the fast path for the same computation."""
        self._check_mode(mode)
        noisy = self.scheduler.add_noise(original_samples=inputs, noise=noise, timesteps=timesteps)
        if mode == "concat" and condition is not None:  # un-noised condition channels behind the noised ones; no cross-attention context
            noisy, condition = torch.cat([noisy, condition.to(noisy.dtype)], dim=1), None
        return diffusion_model(x=noisy, timesteps=timesteps, context=condition)

    @torch.no_grad()
    def sample(self, input_noise, diffusion_model, scheduler=None, save_intermediates=False, intermediate_steps=100, conditioning=None,
               mode="crossattn", verbose=True, noises=None, generator=None, use_graph=True):
        """conditioning with mode="concat": fp32 NC'[D]HW, model_input = cat([image, conditioning], dim=1) at every step; with
        mode="crossattn": fp32 [N, tokens, cross_attention_dim], the model's `context` (train_ldm.py:349-365 passes neither).
        noises: optional sequence with the z of every step (tests pin them); generator: torch generator for the per-step noise."""
        self._check_mode(mode)
        if not input_noise.is_cuda:
            raise RuntimeError("medical_image_generation_amd runs on MI355X only: move the inputs to 'cuda'")
        cc, ctx_shape = 0, None
        if conditioning is not None:
            conditioning = conditioning.to(device=input_noise.device, dtype=F32).contiguous()
            if mode == "concat":
                if conditioning.dim() != input_noise.dim() or conditioning.shape[0] != input_noise.shape[0] or \
                        conditioning.shape[2:] != input_noise.shape[2:]:
                    raise ValueError("mode='concat': conditioning must have the sample's batch and spatial shape")
                cc = conditioning.shape[1]
            else:
                if not getattr(diffusion_model, "with_conditioning", False):
                    raise ValueError("model should have with_conditioning = True if context is provided")
                if conditioning.dim() != 3 or conditioning.shape[0] != input_noise.shape[0] or \
                        conditioning.shape[2] != diffusion_model.cross_attention_dim:
                    raise ValueError(f"mode='crossattn': conditioning must be [batch, tokens, {diffusion_model.cross_attention_dim}]")
                ctx_shape = (conditioning.shape[1], conditioning.shape[2])
        elif getattr(diffusion_model, "with_conditioning", False):
            raise ValueError("the model was built with_conditioning=True: pass conditioning=[batch, tokens, cross_attention_dim]")
        if input_noise.shape[1] + cc != diffusion_model.in_channels:
            raise ValueError(f"Input number of channels ({input_noise.shape[1] + cc}) is not equal to expected number of channels "
                             f"({diffusion_model.in_channels})")
        scheduler = scheduler or self.scheduler
        inter = []

        def on_step(i, t, x):
            if save_intermediates and t % intermediate_steps == 0:
                inter.append(x.clone())

        loop = self._loop(diffusion_model, scheduler, input_noise.shape, input_noise.device, cc, ctx_shape)
        image = loop.run(input_noise.float(), noises=noises, generator=generator, use_graph=use_graph, on_step=on_step, conditioning=conditioning)
        return (image, inter) if save_intermediates else image


class LatentDiffusionInferer(DiffusionInferer):
    """`generative.inferers.LatentDiffusionInferer(scheduler, scale_factor)`: samples latents, then decodes
    `autoencoder_model.decode_stage_2_outputs(latents / scale_factor)` (train_ldm.py:112, 362-364)."""

    def __init__(self, scheduler, scale_factor=1.0):
        super().__init__(scheduler)
        self.scale_factor = scale_factor

    def __call__(self, inputs, autoencoder_model, diffusion_model, noise, timesteps, condition=None, mode="crossattn"):
        """`generative.inferers.LatentDiffusionInferer.__call__`: encode (no grad) -> * scale_factor -> DiffusionInferer.__call__."""
        with torch.no_grad():
            latent = autoencoder_model.encode_stage_2_inputs(inputs) * self.scale_factor
        return super().__call__(inputs=latent, diffusion_model=diffusion_model, noise=noise, timesteps=timesteps, condition=condition,
                                mode=mode)

    @torch.no_grad()
    def sample(self, input_noise, autoencoder_model, diffusion_model, scheduler=None, save_intermediates=False, intermediate_steps=100,
               conditioning=None, mode="crossattn", verbose=True, noises=None, generator=None, use_graph=True):
        out = super().sample(input_noise, diffusion_model, scheduler, save_intermediates, intermediate_steps, conditioning, mode, verbose,
                             noises=noises, generator=generator, use_graph=use_graph)
        latent, inter = out if save_intermediates else (out, None)
        image = autoencoder_model.decode_stage_2_outputs(latent / self.scale_factor)
        if save_intermediates:
            return image, [autoencoder_model.decode_stage_2_outputs(z / self.scale_factor) for z in inter]
        return image
