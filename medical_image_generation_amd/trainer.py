"""Fused DDPM train step on the HIP path (the hot loop of train_ldm.LDM.train_one_epoch, T-LDM:132-191, and of
train_ddpm.DDPM.train_one_epoch, T-DDPM:175-209):

    noise -> q-sample -> DiffusionModelUNet -> MSE -> backward -> (DDP all-reduce) -> clip_grad_norm_ -> Adam[W]

No torch autograd and no torch compute kernels in the step: the tape engine drives our kernels directly, the
optimizer is one fused launch over the flat parameter arena, and the whole step can be captured in a hipGraph
(`capture=True`) and replayed, which removes the Python / launch overhead of ~600 kernel launches.

`AETrainer` is the same thing for the generator step of train_autoencoder.AutoEncoder.train_one_epoch (T-AE:406-435)
without its third-party perceptual/adversarial terms: encode -> sample -> decode -> L1 + kl_weight * KL -> backward -> Adam.

Data parallelism: one process per GPU; the trainable prefix of the flat gradient arena is all-reduced
(average) over RCCL in a few large buckets -- statically unused `proj_attn.*` tensors live outside that prefix.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ddp
from . import engine as E
from . import hipops as ops
from ._lib import call, ptr

F32 = torch.float32


class DDPMSchedule:
    """alphas_cumprod of generative's DDPMScheduler (closed form; see oracle/step.py for provenance)."""

    def __init__(self, num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205,
                 prediction_type="epsilon", device="cuda"):
        if schedule == "scaled_linear_beta":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=F32) ** 2
        elif schedule == "linear_beta":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=F32)
        else:
            raise ValueError(f"unknown schedule {schedule}")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"unknown prediction_type {prediction_type} (train_ldm.py:163-166 knows 'epsilon' and 'v_prediction')")
        acp = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps, self.prediction_type = num_train_timesteps, prediction_type
        self.sqrt_acp = acp.sqrt().to(device)
        self.sqrt_1macp = (1.0 - acp).sqrt().to(device)


class _ArenaTrainer:
    """Optimizer state over the flat arena, gradient averaging and hipGraph capture; subclasses give forward_backward()."""

    def __init__(self, model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device):
        self.model = model
        self.device = torch.device(device or "cuda")
        self.lr, self.betas, self.eps = lr, betas, eps
        self.decoupled = optimizer == "AdamW"
        if optimizer not in ("AdamW", "Adam"):
            raise ValueError("optimizer must be 'Adam' or 'AdamW'")
        self.weight_decay = (0.01 if self.decoupled else 0.0) if weight_decay is None else weight_decay  # torch defaults
        self.max_grad_norm = max_grad_norm
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        self.bucket_elems = bucket_mb * (1 << 20) // 4
        self.arena = model.arena(self.device)
        if self.world > 1:
            ddp.broadcast_parameters(self.arena.data, 0, process_group)
        n = self.arena.n_trainable
        self.exp_avg = torch.zeros(n, dtype=F32, device=self.device)
        self.exp_avg_sq = torch.zeros(n, dtype=F32, device=self.device)
        self.step_count = torch.zeros(1, dtype=F32, device=self.device)
        self.sumsq = torch.zeros(1, dtype=F32, device=self.device)
        self.loss = torch.zeros(1, dtype=F32, device=self.device)
        self._graph = None
        self._static = None

    # ------------------------------------------------------------------ pieces
    def all_reduce_grads(self):
        if self.world > 1:
            ddp.average_gradients(self.arena.grad, self.arena.n_trainable, self.pg, self.bucket_elems)

    def optimizer_step(self):
        a = self.arena
        n = a.n_trainable
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            call("mi_sumsq_f32", ptr(a.grad), n, ptr(self.sumsq), 0)
        call("mi_adam_step", ptr(a.data), ptr(a.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), n, self.lr, self.betas[0], self.betas[1],
             self.eps, self.weight_decay, int(self.decoupled), ptr(self.sumsq) if clip else None, float(self.max_grad_norm or 0.0),
             ptr(self.step_count))

    # ------------------------------------------------------------------ one step
    def step(self, *inputs):
        """Eager step; returns the (device) loss tensor without synchronising."""
        self.forward_backward(*inputs)
        self.all_reduce_grads()
        self.optimizer_step()
        return self.loss

    def capture(self, *inputs, warmup=2):
        """Capture forward+backward and the optimizer as hipGraphs around static input buffers (the all-reduce stays
        eager between them so RCCL is never inside a capture).  Call step_graph() afterwards."""
        self._static = tuple(t.clone() for t in inputs)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):  # creates conv plans / workspaces (hipMalloc) outside capture
                self.forward_backward(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # thread_local: a collective library's watchdog thread polling events must not invalidate the capture
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
            self.forward_backward(*self._static)
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, capture_error_mode="thread_local"):
            self.optimizer_step()
        self._graph = True

    def step_graph(self, *inputs):
        """Replay; inputs given (positionally, None = keep) are copied into the static buffers first."""
        assert self._graph, "call capture() first"
        for buf, t in zip(self._static, inputs):
            if t is not None:
                buf.copy_(t)
        self._g_fb.replay()
        self.all_reduce_grads()
        self._g_opt.replay()
        return self.loss


class DDPMTrainer(_ArenaTrainer):
    """step(x0, noise, timesteps[, class_labels]): x0/noise fp32 NCDHW, timesteps (and class_labels) int64 [N]."""

    def __init__(self, model, lr=2e-5, optimizer="AdamW", weight_decay=None, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 schedule: DDPMSchedule | None = None, process_group=None, bucket_mb=64, device=None):
        super().__init__(model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device)
        self.schedule = schedule or DDPMSchedule(device=self.device)

    def forward_backward(self, x0, noise, timesteps, class_labels=None):
        """q-sample -> UNet -> MSE -> backward into the gradient arena.  x0/noise: fp32 NCDHW, timesteps: int64 [N];
        class_labels: int64 [N], only for a net built with num_class_embeds."""
        m = self.model
        a = self.arena
        a.grad.zero_()
        sd = m.spatial_dims
        n, c = x0.shape[0], x0.shape[1]
        sp = tuple(x0.shape[2:])
        v = 1
        for s in sp:
            v *= s
        dims = (1,) * (3 - sd) + sp
        x_t = torch.empty((n,) + dims + (c,), dtype=torch.bfloat16, device=x0.device)
        vpred = self.schedule.prediction_type == "v_prediction"  # target = scheduler.get_velocity(x0, noise, t), T-LDM:163-165
        target = torch.empty_like(noise) if vpred else noise
        call("mi_qsample", ptr(x0), ptr(noise), ptr(self.schedule.sqrt_acp), ptr(self.schedule.sqrt_1macp), ptr(timesteps), ptr(x_t),
             ptr(target) if vpred else None, n, c, v)
        ctx = E.Ctx(a, m._plans, grad_enabled=True, prepacked=m.pack_all())
        if (class_labels is None) != (getattr(m, "num_class_embeds", None) is None):
            raise ValueError("class_labels should be provided exactly when the model has num_class_embeds")
        pred = m._run(ctx, x_t, timesteps, need_dx=False, class_labels=class_labels)
        dpred = torch.empty_like(pred)
        call("mi_mse_fwd_bwd", ptr(pred), ptr(target), ptr(dpred), ptr(self.loss), n, pred.shape[-1], v, 1.0)
        ctx.tape.backward(pred, dpred)
        ctx.tape.grads.clear(), ctx.tape.keep.clear()


class AETrainer(_ArenaTrainer):
    """step(images, eps): images fp32 NCDHW, eps fp32 latent-shaped NCDHW (the torch.randn_like of AEKL:786-787 made an
    input so that a captured graph sees fresh noise and tests can pin it).  Defaults are the reference's generator
    optimizer (Adam, lr 5e-5, grad_clip_max_norm 1; T-AE:428-434, 470) and 3-D kl_weight (CFG:995-1026)."""

    def __init__(self, model, lr=5e-5, optimizer="Adam", weight_decay=None, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 kl_weight=1e-7, process_group=None, bucket_mb=64, device=None):
        super().__init__(model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device)
        self.kl_weight = float(kl_weight)

    def forward_backward(self, images, eps):
        m, a = self.model, self.arena
        a.grad.zero_()
        self.loss.zero_()
        n, c = images.shape[0], images.shape[1]
        sp = tuple(images.shape[2:])
        v = 1
        for s in sp:
            v *= s
        x_cl = ops.to_channels_last(images)
        ctx = E.Ctx(a, m._plans, grad_enabled=True, prepacked=m.pack_all())
        tape = ctx.tape
        mu, sigma = m._encode_run(ctx, x_cl, need_dx=False)
        lc = mu.shape[-1]
        lv = mu.numel() // (n * lc)
        z = torch.empty_like(mu)
        call("mi_reparam_kl_fwd", ptr(mu), ptr(sigma), ptr(eps), ptr(z), ptr(self.loss), n, lc, lv, self.kl_weight)

        def bwd():
            dz = tape.take(z)
            dmu, dsigma = torch.empty_like(mu), torch.empty_like(sigma)
            call("mi_reparam_kl_bwd", ptr(mu), ptr(sigma), ptr(eps), ptr(dz), ptr(dmu), ptr(dsigma), n, lc, lv, self.kl_weight)
            tape.put(mu, dmu), tape.put(sigma, dsigma)

        tape.record(bwd)
        recon = m._decode_run(ctx, z, True)
        drecon = torch.empty_like(recon)
        call("mi_l1_fwd_bwd", ptr(recon), ptr(images), ptr(drecon), ptr(self.loss), n, recon.shape[-1], v, 1)
        tape.backward(recon, drecon)
        tape.grads.clear(), tape.keep.clear()
