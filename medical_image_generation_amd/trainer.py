"""Fused DDPM train step on the HIP path (the hot loop of train_ldm.LDM.train_one_epoch, T-LDM:132-191, and of
train_ddpm.DDPM.train_one_epoch, T-DDPM:175-209):

    noise -> q-sample -> DiffusionModelUNet -> MSE -> backward -> (DDP all-reduce) -> clip_grad_norm_ -> Adam[W]

No torch autograd and no torch compute kernels in the step: the tape engine drives our kernels directly, the
optimizer is one fused launch over the flat parameter arena, and the whole step can be captured in a hipGraph
(`capture=True`) and replayed, which removes the Python / launch overhead of ~600 kernel launches.

`AETrainer` is the same thing for the generator step of train_autoencoder.AutoEncoder.train_one_epoch (T-AE:406-435):
encode -> sample -> decode -> L1 + kl_weight * KL (+ the caller's perceptual / adversarial terms through `extra_loss`, evaluated by
torch autograd on the reconstruction: those networks are third-party torch modules) -> backward -> Adam.

Data parallelism: one process per GPU; the trainable prefix of the flat gradient arena is all-reduced
(average) over RCCL in a few large buckets -- statically unused `proj_attn.*` tensors live outside that prefix.
"""
from __future__ import annotations

import os

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
    """Optimizer state over the flat arena, gradient accumulation, data-parallel gradient exchange and hipGraph capture;
    subclasses give forward_backward().

    Data parallelism: the backward pass is cut ONCE, at the point where every gradient of the arena's *early* segment
    [arena.n_late, arena.n_trainable) is final (HipModule orders the arena so: the parameters whose gradients complete last -- the
    finest-resolution levels, conv_in and everything the time-embedding backward writes -- form the *late* prefix).  The SUM
    all-reduce of the early segment (>= 95 % of the bytes) is issued asynchronously at the cut and runs over xGMI while the rest of
    the backward (the full-resolution layers: ~1/3 of its time, almost no parameters) computes; the small late segment follows.
    The mean is never materialised: 1/world is folded into the optimizer kernel (`grad_scale`).

    overlap=False (or MI_DDP_OVERLAP=0) is the plain schedule: ONE backward (one hipGraph), then the all-reduce of the whole
    trainable prefix, then the optimizer -- the fallback should the overlapped exchange misbehave on a given RCCL / node (the N > 1
    overlap has been rehearsed with gloo ranks and with one-rank RCCL groups only: DESIGN.md section 6)."""

    def __init__(self, model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device,
                 grad_accumulate_step=1, overlap=None):
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
        if int(grad_accumulate_step) < 1:
            raise ValueError("grad_accumulate_step must be >= 1")
        self.grad_accumulate_step = int(grad_accumulate_step)
        self.arena = model.arena(self.device)
        if self.world > 1:
            ddp.broadcast_parameters(self.arena.data, 0, process_group)
        n = self.arena.n_trainable
        self.exp_avg = torch.zeros(n, dtype=F32, device=self.device)
        self.exp_avg_sq = torch.zeros(n, dtype=F32, device=self.device)
        self.step_count = torch.zeros(1, dtype=F32, device=self.device)
        self.sumsq = torch.zeros(1, dtype=F32, device=self.device)
        self.loss = torch.zeros(1, dtype=F32, device=self.device)
        self._accum = None       # fp32 [n_trainable]: sum of the gradients of the pending micro-steps (grad_accumulate_step > 1)
        self._micro = 0          # micro-steps since the last optimizer step
        self._graph = None
        self._static = None
        self._force_split = False  # tests: capture the two-graph form with world 1
        self._force_exchange = False  # tests: issue the (one-rank) collectives although world == 1
        self.overlap = (os.environ.get("MI_DDP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)

    # ------------------------------------------------------------------ pieces
    def _forward(self, *inputs):
        """-> (tape, out, dout): the network forward and the loss gradient at its output (subclasses)."""
        raise NotImplementedError

    def _fb_begin(self, *inputs):
        """Forward + the backward up to the model's cut mark (engine.CUT; the whole backward when nothing was marked).  Returns the
        state _fb_finish needs; it also keeps every tensor that is live across the cut alive, so the two halves can be captured
        into two hipGraphs sharing one memory pool."""
        tape, out, dout = self._forward(*inputs)
        tape.put(out, dout)
        fns, tape.fns = tape.fns, []
        i = len(fns)
        while i > 0:
            i -= 1
            if fns[i] is E.CUT:
                break
            fns[i]()
        E.join_side(self.device)  # (the cut ends a hipGraph / starts the early all-reduce: nothing may still be running beside it)
        return tape, fns[:i]

    def _fb_finish(self, state):
        tape, rest = state
        for fn in reversed(rest):
            fn()  # (a second CUT mark is a no-op)
        E.join_side(self.device)
        tape.grads.clear(), tape.keep.clear()

    def forward_backward(self, *inputs, on_cut=None):
        """Fresh gradients of one (micro-)batch into arena.grad, loss into self.loss.  on_cut(): called where every gradient of the
        arena's early segment [n_late, n_trainable) is final (data-parallel overlap)."""
        state = self._fb_begin(*inputs)
        if on_cut is not None:
            on_cut()
        self._fb_finish(state)

    def _exchanging(self):
        return self.world > 1 or self._force_exchange

    def _exchange(self):
        return ddp.GradientExchange(self.arena.grad, self.arena.n_late, self.arena.n_trainable, self.pg, self.bucket_elems,
                                    force=self._force_exchange)

    def all_reduce_grads(self):
        """SUM all-reduce of the whole trainable gradient prefix, no overlap (the mean is taken by the optimizer's grad_scale)."""
        if self._exchanging():
            self._exchange().finish()

    def optimizer_step(self):
        a = self.arena
        n = a.n_trainable
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            call("mi_sumsq_f32", ptr(a.grad), n, ptr(self.sumsq), 0)
        call("mi_adam_step", ptr(a.data), ptr(a.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), n, self.lr, self.betas[0], self.betas[1],
             self.eps, self.weight_decay, int(self.decoupled), ptr(self.sumsq) if clip else None, float(self.max_grad_norm or 0.0),
             1.0 / self.world, ptr(self.step_count))

    # ------------------------------------------------------------------ gradient accumulation (T-LDM:173-180, T-AE:389-397)
    def _fold_micro_step(self, boundary):
        """The reference sums the micro-step gradients (loss NOT divided) and clips / steps / zeroes on the boundary.  Every
        forward_backward() writes a fresh gradient into arena.grad; pending micro-steps live in a side buffer, so kernels that
        COPY gradients between tensors (a shortcut conv's bias gradient) never see stale sums."""
        n = self.arena.n_trainable
        if not boundary:
            if self._accum is None:
                self._accum = torch.empty(n, dtype=F32, device=self.device)
                call("mi_zero_f32_2d", ptr(self._accum), n, 1, n)
            call("mi_axpy_f32", ptr(self._accum), ptr(self.arena.grad), 1.0, n)
            self._micro += 1
            return
        if self._micro:
            call("mi_axpy_f32", ptr(self.arena.grad), ptr(self._accum), 1.0, n)
            call("mi_zero_f32_2d", ptr(self._accum), n, 1, n)
        self._micro = 0

    # ------------------------------------------------------------------ one step
    def step(self, *inputs, last_in_epoch=False):
        """One (micro-)step, eager; returns the (device) loss tensor of this micro-batch without synchronising.  With
        grad_accumulate_step = k the optimizer runs on every k-th call (or when last_in_epoch, T-LDM:173); the data-parallel
        exchange happens on that boundary micro-step only."""
        boundary = (self._micro + 1) % self.grad_accumulate_step == 0 or last_in_epoch
        ex = self._exchange() if self._exchanging() and boundary else None
        # the early segment may leave at the cut only when no pending micro-step sum has to be folded in first
        self.forward_backward(*inputs, on_cut=ex.start_early if ex is not None and self._micro == 0 and self.overlap else None)
        self._fold_micro_step(boundary)
        if not boundary:
            return self.loss
        if ex is not None:
            ex.finish()
        self.optimizer_step()
        return self.loss

    def capture(self, *inputs, warmup=2):
        """Capture the step as hipGraphs around static input buffers; call step_graph() afterwards.  Graphs: forward + the backward up
        to the cut, the rest of the backward (only when there is something to overlap: world > 1), the optimizer.  RCCL is never
        inside a capture: the all-reduces are issued between the replays, the early segment's concurrently with the second graph."""
        if self.grad_accumulate_step != 1:
            raise RuntimeError("hipGraph capture covers grad_accumulate_step == 1; use step() for accumulation")
        self._static = tuple(None if t is None else t.clone() for t in inputs)  # (None: an optional input that is not used)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):  # creates conv plans / workspaces (hipMalloc) outside capture
                self.forward_backward(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        split = self.overlap and (self.world > 1 or self._force_split) and 0 < self.arena.n_late < self.arena.n_trainable
        # thread_local: a collective library's watchdog thread polling events must not invalidate the capture
        self._g_fb, self._g_fb2 = torch.cuda.CUDAGraph(), None
        if not split:
            with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
                self.forward_backward(*self._static)
        else:
            with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
                state = self._fb_begin(*self._static)
            self._g_fb2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_fb2, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
                self._fb_finish(state)
            del state
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, capture_error_mode="thread_local"):
            self.optimizer_step()
        self._graph = True
        # The graphs hold raw device pointers of objects Python owns: the PackBatch (group tables), every conv plan (packed weights,
        # split slabs), the GroupNorm workspace and the arena.  Pin them for as long as the graphs exist -- a forward at another
        # shape between train steps (validation, DiffusionInferer.sample) replaces model._packb / grows the workspace, and their
        # destructors would hipFree memory the next replay still writes to.
        self._pinned = [(m._packb, dict(m._plans), m._arena) for m in self._models()] + [dict(ops._ws_cache)]

    def _models(self):
        return [self.model]

    def step_graph(self, *inputs):
        """Replay; inputs given (positionally, None = keep) are copied into the static buffers first."""
        assert self._graph, "call capture() first"
        if self.model._arena is not self.arena:  # .to() / .cuda() / load into new storage: the graphs point at the old buffers
            raise RuntimeError("the model's parameter arena was rebuilt after capture(): create a new trainer (or call capture() again)")
        for buf, t in zip(self._static, inputs):
            if t is not None:
                if buf is None:
                    raise ValueError("this input was None at capture(): capture again with a tensor in its place")
                buf.copy_(t)
        ex = self._exchange() if self._exchanging() else None
        self._g_fb.replay()
        if self._g_fb2 is not None:
            if ex is not None:
                ex.start_early()   # early segment: final once the first graph is done ...
            self._g_fb2.replay()   # ... and its all-reduce runs over xGMI while the rest of the backward computes
        if ex is not None:
            ex.finish()
        self._g_opt.replay()
        return self.loss


class DDPMTrainer(_ArenaTrainer):
    """step(x0, noise, timesteps[, class_labels][, context=...][, condition=...]): x0/noise fp32 NCDHW, timesteps (and class_labels)
    int64 [N]; context: fp32 [N, tokens, cross_attention_dim] for a net built with with_conditioning=True (the `context=` of
    DiffusionModelUNet.forward, UNet:1936-1944), a constant of the step like in forward().

    condition: channel-concatenation conditioning (the third-party inferer's `condition=..., mode="concat"`, the arguments of the call at
    train_ddpm.py:191; BASELINE configs[4] "label-channel conditioning"): fp32 NC'[D]HW, spatially like x0.  Only x0's channels are
    noised; the condition channels are written un-noised behind them -- model input = cat([noisy, condition], dim=1) with
    in_channels = C + C' -- and the model predicts x0's C channels (out_channels = C), compared with x0's noise."""

    def __init__(self, model, lr=2e-5, optimizer="AdamW", weight_decay=None, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 schedule: DDPMSchedule | None = None, process_group=None, bucket_mb=64, device=None, grad_accumulate_step=1, overlap=None):
        super().__init__(model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device,
                         grad_accumulate_step, overlap)
        self.schedule = schedule or DDPMSchedule(device=self.device)

    def _forward(self, x0, noise, timesteps, class_labels=None, context=None, condition=None):
        """q-sample -> UNet -> MSE (+ its gradient).  x0/noise: fp32 NCDHW, timesteps: int64 [N]; class_labels: int64 [N], only for
        a net built with num_class_embeds; context: fp32 [N, tokens, cross_attention_dim], only for with_conditioning=True;
        condition: fp32 NC'[D]HW concatenated un-noised behind the noised channels (mode="concat")."""
        m = self.model
        a = self.arena
        # raw pointers go to the kernels: refuse anything they would misread instead of reading out of bounds
        if not (x0.is_cuda and noise.is_cuda and timesteps.is_cuda):
            raise RuntimeError("DDPMTrainer inputs must live on the GPU")
        if x0.dtype != F32 or noise.dtype != F32 or not x0.is_contiguous() or not noise.is_contiguous() or x0.shape != noise.shape:
            raise ValueError("x0 and noise must be contiguous fp32 tensors of the same NC[D]HW shape")
        if timesteps.dtype != torch.int64 or timesteps.shape != (x0.shape[0],) or not timesteps.is_contiguous():
            raise ValueError("timesteps must be a contiguous int64 tensor of shape [N]")  # range: clamped to the schedule in k_qsample
        cc = 0
        if condition is not None:
            if not condition.is_cuda or condition.dtype != F32 or not condition.is_contiguous() or condition.dim() != x0.dim() or \
                    condition.shape[0] != x0.shape[0] or condition.shape[2:] != x0.shape[2:]:
                raise ValueError("condition must be a contiguous fp32 GPU tensor with x0's batch and spatial shape (mode='concat')")
            cc = condition.shape[1]
        if x0.shape[1] + cc != m.in_channels:
            raise ValueError(f"Input number of channels ({x0.shape[1] + cc}) is not equal to expected number of channels ({m.in_channels})")
        if x0.shape[1] != m.out_channels:  # the MSE compares the prediction with x0's noise channel for channel
            raise ValueError(f"the model predicts {m.out_channels} channels but the noised input has {x0.shape[1]}: the loss target "
                             "(the noise / velocity of x0) must have the prediction's channel count; un-noised conditioning channels go in "
                             "`condition=`")
        a.grad.zero_()
        sd = m.spatial_dims
        n, c = x0.shape[0], x0.shape[1]
        sp = tuple(x0.shape[2:])
        v = 1
        for s in sp:
            v *= s
        dims = (1,) * (3 - sd) + sp
        x_t = torch.empty((n,) + dims + (c + cc,), dtype=torch.bfloat16, device=x0.device)
        vpred = self.schedule.prediction_type == "v_prediction"  # target = scheduler.get_velocity(x0, noise, t), T-LDM:163-165
        target = torch.empty_like(noise) if vpred else noise
        call("mi_qsample", ptr(x0), ptr(noise), ptr(self.schedule.sqrt_acp), ptr(self.schedule.sqrt_1macp), ptr(timesteps), ptr(condition), cc,
             ptr(x_t), ptr(target) if vpred else None, n, c, v, self.schedule.num_train_timesteps)
        ctx = E.Ctx(a, m._plans, grad_enabled=True, prepacked=m.pack_all())
        if (class_labels is None) != (getattr(m, "num_class_embeds", None) is None):
            raise ValueError("class_labels should be provided exactly when the model has num_class_embeds")
        ctx_tokens = None
        if (context is not None) != bool(getattr(m, "with_conditioning", False)):
            raise ValueError("context should be provided exactly when the model has with_conditioning = True")
        if context is not None:
            if not context.is_cuda or context.dtype != F32 or context.dim() != 3 or context.shape[0] != n or context.shape[2] != m.cross_attention_dim:
                raise ValueError(f"context must be a GPU fp32 tensor [batch, tokens, {m.cross_attention_dim}]")
            ctx_tokens = ops.cast_bf16(context.contiguous().reshape(-1, context.shape[2]))
        pred = m._run(ctx, x_t, timesteps, need_dx=False, class_labels=class_labels, context=ctx_tokens)
        dpred = torch.empty_like(pred)
        if pred.shape[-1] != target.shape[1]:  # k_mse indexes both with ONE channel count
            raise ValueError(f"prediction has {pred.shape[-1]} channels, the target {target.shape[1]}")
        call("mi_mse_fwd_bwd", ptr(pred), ptr(target), ptr(dpred), ptr(self.loss), n, pred.shape[-1], v, 1.0)
        self._keep = (x_t, target, ctx_tokens, condition)  # read by kernels still in flight / by the second graph of a split capture
        return ctx.tape, pred, dpred


class LDMTrainer(DDPMTrainer):
    """The latent-diffusion train step as train_ldm.LDM.train_one_epoch runs it (T-LDM:145-180, 'vae' branch):

        latents = autoencoder.encode_stage_2_inputs(images)   (no grad: T-LDM:154-156)  ->  * scale_factor  ->  q-sample -> UNet
        -> MSE -> backward -> clip -> AdamW

    step(images, eps, noise, timesteps[, class_labels][, context=][, condition=]): images fp32 NC[D]HW; eps fp32, latent-shaped -- the torch.randn_like of
    AutoencoderKL.sampling (AEKL:786-787) made an input so that a captured graph sees fresh noise and tests can pin it; noise fp32,
    latent-shaped (T-LDM:159); timesteps int64 [N].  The encoder runs tape-less on the same HIP kernels inside the same hipGraph.
    `scale_factor`: 1 / std of the first batch's latents (T-LDM:110-112) -- `estimate_scale_factor` -- or given."""

    def __init__(self, model, autoencoder, scale_factor: float | None = None, **kw):
        super().__init__(model, **kw)
        self.autoencoder = autoencoder
        self.ae_arena = autoencoder.arena(self.device)
        self.scale_factor = None if scale_factor is None else float(scale_factor)

    def _models(self):
        return [self.model, self.autoencoder]

    def _latents(self, images, eps):
        """encode_stage_2_inputs (AEKL:827-830): z = z_mu + eps * z_sigma, fp32 NC[D]HW, unscaled; no tape."""
        ae = self.autoencoder
        if images.dtype != F32 or not images.is_cuda or not images.is_contiguous() or eps.dtype != F32 or not eps.is_contiguous():
            raise ValueError("images and eps must be contiguous fp32 GPU tensors")
        x_cl = ops.to_channels_last(images)
        ctx = E.Ctx(self.ae_arena, ae._plans, grad_enabled=False, prepacked=ae.pack_all())
        mu, sigma = ae._encode_run(ctx, x_cl, need_dx=False)
        n, lc = mu.shape[0], mu.shape[-1]
        lv = mu.numel() // (n * lc)
        if tuple(eps.shape) != (n, lc) + tuple(d for d in mu.shape[1:4])[3 - ae.spatial_dims:]:
            raise ValueError(f"eps must have the latent shape, got {tuple(eps.shape)}")
        z = torch.empty_like(mu)
        call("mi_reparam_kl_fwd", ptr(mu), ptr(sigma), ptr(eps), ptr(z), None, n, lc, lv, 0.0)
        return ops.to_channels_first(z, ae.spatial_dims)

    @torch.no_grad()
    def estimate_scale_factor(self, images, eps):
        """scale_factor = 1 / std(z) over the first batch (T-LDM:110-112); stored and returned."""
        self.scale_factor = float(1.0 / torch.std(self._latents(images, eps)))
        return self.scale_factor

    def _forward(self, images, eps, noise, timesteps, class_labels=None, context=None, condition=None):
        """condition: latent-shaped fp32 tensor (e.g. the label mask resampled to the latent grid) concatenated un-noised behind the
        scaled latents -- BASELINE configs[4]."""
        if self.scale_factor is None:
            raise RuntimeError("scale_factor is not set: pass it or call estimate_scale_factor(first_batch, eps) (train_ldm.py:110-112)")
        z = self._latents(images, eps)
        call("mi_scale_f32", ptr(z), self.scale_factor, z.numel())  # latents_scaled = latents * inferer.scale_factor (T-LDM:157)
        return super()._forward(z, noise, timesteps, class_labels, context, condition)


class AETrainer(_ArenaTrainer):
    """step(images, eps): images fp32 NCDHW, eps fp32 latent-shaped NCDHW (the torch.randn_like of AEKL:786-787 made an
    input so that a captured graph sees fresh noise and tests can pin it).  Defaults are the reference's generator
    optimizer (Adam, lr 5e-5, grad_clip_max_norm 1; T-AE:428-434, 470) and 3-D kl_weight (CFG:995-1026).

    extra_loss: the generator step's third-party terms (T-AE:411-421: `perceptual_loss(recon, images) * perc_weight` and, after the
    warm-up epochs, `adv_loss(discriminator(recon)[-1], target_is_real=True, for_discriminator=False) * adv_weight`).  A callable
    (reconstruction fp32 NCDHW with requires_grad, images) -> scalar torch loss; it runs under torch autograd on the GPU and its
    gradient with respect to the reconstruction joins the L1 gradient before the HIP backward.  The networks inside it
    (`generative`'s PatchDiscriminator / PerceptualLoss with downloaded weights) stay the user's torch modules -- they are not
    rebuilt here.  `reconstruction` (fp32 NCDHW, detached) holds the last step's output for the caller's discriminator step
    (T-AE:371-397), which is plain torch on the caller's side."""

    def __init__(self, model, lr=5e-5, optimizer="Adam", weight_decay=None, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0,
                 kl_weight=1e-7, process_group=None, bucket_mb=64, device=None, grad_accumulate_step=1, extra_loss=None, overlap=None):
        super().__init__(model, lr, optimizer, weight_decay, betas, eps, max_grad_norm, process_group, bucket_mb, device,
                         grad_accumulate_step, overlap)
        self.kl_weight = float(kl_weight)
        self.extra_loss = extra_loss
        self.reconstruction = None
        self.extra_loss_value = None

    def _forward(self, images, eps):
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
        if self.extra_loss is not None:
            rec = ops.to_channels_first(recon, len(sp)).requires_grad_(True)
            with torch.enable_grad():
                extra = self.extra_loss(rec, images)
                (g,) = torch.autograd.grad(extra, rec)
            self.extra_loss_value = extra.detach()
            self.loss += self.extra_loss_value.to(self.loss.dtype).reshape(self.loss.shape)
            drecon = ops.add(drecon, ops.to_channels_last(g))
            self.reconstruction = rec.detach()
        return tape, recon, drecon


class AEGANTrainer(AETrainer):
    """The autoencoder's GAN step with BOTH networks on the HIP path (train_autoencoder.AutoEncoder.train_one_epoch, T-AE:371-435):

        generator      (T-AE:406-435): recon = AE(images); loss_g = L1 + kl_weight * KL [+ extra_loss] + (adversarial: after the warm-up
                       epochs, T-AE:416) adv_weight * LS(D(recon)[-1], real)  ->  backward through D (frozen) and the AE -> clip -> Adam
        discriminator  (T-AE:371-397): loss_d = adv_weight * 0.5 * (LS(D(recon.detach())[-1], fake) + LS(D(images)[-1], real))
                       ->  backward through D -> clip -> Adam (lr 5e-5, T-AE:471)

    `discriminator`: medical_image_generation_amd.discriminator.PatchDiscriminator; LS = PatchAdversarialLoss("least_squares").
    step(images, eps) runs both (the reference's order: generator, then discriminator on the same reconstruction); `adversarial`
    switches the adversarial term and the discriminator step on (epoch >= autoencoder_warm_up_epochs).  The perceptual term needs
    downloaded weights and stays the caller's `extra_loss`.  capture() / step_graph() replay generator and discriminator graphs."""

    def __init__(self, model, discriminator, adv_weight=0.01, d_lr=5e-5, adversarial=True, **kw):
        super().__init__(model, **kw)
        if self.grad_accumulate_step != 1:
            raise NotImplementedError("AEGANTrainer covers grad_accumulate_step == 1")
        from .discriminator import PatchAdversarialLoss
        self.D = discriminator
        self.d_arena = discriminator.arena(self.device)
        if self.world > 1:
            ddp.broadcast_parameters(self.d_arena.data, 0, self.pg)
        self.adv, self.adv_weight, self.d_lr = PatchAdversarialLoss("least_squares"), float(adv_weight), float(d_lr)
        self.adversarial = bool(adversarial)
        n = self.d_arena.n_trainable
        self.d_exp_avg = torch.zeros(n, dtype=F32, device=self.device)
        self.d_exp_avg_sq = torch.zeros(n, dtype=F32, device=self.device)
        self.d_step_count = torch.zeros(1, dtype=F32, device=self.device)
        self.d_sumsq = torch.zeros(1, dtype=F32, device=self.device)
        self.gen_loss = torch.zeros(1, dtype=F32, device=self.device)   # adv_weight * LS(D(recon), real) of the last generator step
        self.disc_loss = torch.zeros(1, dtype=F32, device=self.device)  # adv_weight * 0.5 (fake + real) of the last discriminator step
        self.recon_cl = None
        self._g_d = None

    def _models(self):
        return [self.model]

    def _forward(self, images, eps):
        tape, recon, drecon = super()._forward(images, eps)
        self.recon_cl = recon
        if self.adversarial:
            self.gen_loss.zero_()
            c = E.Ctx(self.d_arena, {}, grad_enabled=True)
            logits = self.D._run(c, recon, need_dx=True, param_grads=False)[-1]  # D frozen: `requires_grad = False` (T-AE:401-402)
            dl = self.adv.hip(logits, True, self.gen_loss, self.adv_weight)
            c.tape.backward(logits, dl)
            g = c.tape.take(recon)
            c.tape.grads.clear(), c.tape.keep.clear()
            drecon = ops.add(drecon, g)
            ops.add_f32_(self.loss.view(1, 1), self.gen_loss.view(1, 1))
        return tape, recon, drecon

    # ---- discriminator step
    def d_forward_backward(self, images):
        """Gradients of loss_d into d_arena.grad (fresh), loss into self.disc_loss; uses the reconstruction of the last generator step."""
        if self.recon_cl is None:
            raise RuntimeError("run the generator step first (the discriminator step scores ITS reconstruction, T-AE:376)")
        self.d_arena.grad.zero_()
        self.disc_loss.zero_()
        for x_cl, real in ((self.recon_cl, False), (ops.to_channels_last(images), True)):
            c = E.Ctx(self.d_arena, {}, grad_enabled=True)
            logits = self.D._run(c, x_cl, need_dx=False)[-1]
            dl = self.adv.hip(logits, real, self.disc_loss, 0.5 * self.adv_weight)
            c.tape.backward(logits, dl)
            c.tape.grads.clear(), c.tape.keep.clear()

    def d_optimizer_step(self):
        a = self.d_arena
        n = a.n_trainable
        if self.world > 1:
            ddp.GradientExchange(a.grad, 0, n, self.pg, self.bucket_elems).finish()
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            call("mi_sumsq_f32", ptr(a.grad), n, ptr(self.d_sumsq), 0)
        call("mi_adam_step", ptr(a.data), ptr(a.grad), ptr(self.d_exp_avg), ptr(self.d_exp_avg_sq), n, self.d_lr, self.betas[0], self.betas[1],
             self.eps, 0.0, 0, ptr(self.d_sumsq) if clip else None, float(self.max_grad_norm or 0.0), 1.0 / self.world, ptr(self.d_step_count))

    def step(self, images, eps, last_in_epoch=False):
        loss = super().step(images, eps)
        if self.adversarial:
            self.d_forward_backward(images)
            self.d_optimizer_step()
        return loss

    def capture(self, images, eps, warmup=2):
        super().capture(images, eps, warmup=warmup)
        if self.adversarial:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.d_forward_backward(self._static[0])
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._g_d = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_d, pool=self._g_fb.pool(), capture_error_mode="thread_local"):
                self.d_forward_backward(self._static[0])
                if self.world <= 1:
                    self.d_optimizer_step()
            self._pinned.append((self.d_arena, dict(ops._ws_cache)))

    def step_graph(self, images=None, eps=None):
        loss = super().step_graph(images, eps)
        if self._g_d is not None:
            self._g_d.replay()
            if self.world > 1:
                self.d_optimizer_step()
        return loss
