"""AutoencoderKL "with strides" on the MI355X HIP path.

Drop-in for the reference class (medimgen/autoencoderkl_with_strides.py:625-834): same constructor keywords
(`downsample_parameters` / `upsample_parameters` = per-level (stride, kernel, padding), possibly per-axis lists),
same methods (`forward`, `encode`, `sampling`, `decode`, `reconstruct`, `encode_stage_2_inputs`,
`decode_stage_2_outputs`), same `state_dict()` names, same ValueErrors (AEKL:672-690).  `encode` and `decode` are each
one autograd edge over the HIP tape engine; `sampling` is the reference's own two-op reparameterisation on the (tiny)
latent tensors.

`use_checkpointing=True` checkpoints the whole encoder and the whole decoder, like the reference's torch.utils.checkpoint calls
(AEKL:761-762, 815-816): their activations are dropped after the forward and recomputed inside the backward (engine.checkpoint;
gradients bit-identical to the stored-activation path).
`use_convtranspose=True` (never set by the reference's planner, CFG:843) runs as the data gradient of the matching strided conv
(engine.conv_transpose) for 3-D kernel-3 / padding-1 / stride-2 (or 1) levels and raises NotImplementedError otherwise.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import engine as E
from . import hipops as ops
from ._lib import call, ptr
from .unet import HipModule, ParamSpec, _axis3, _NetFn, _tuple_rep


class AutoencoderKL(HipModule):
    def __init__(
        self,
        spatial_dims: int,
        in_channels: int = 1,
        out_channels: int = 1,
        num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
        num_channels: Sequence[int] = (32, 64, 64, 64),
        attention_levels: Sequence[bool] = (False, False, True, True),
        latent_channels: int = 3,
        norm_num_groups: int = 32,
        norm_eps: float = 1e-6,
        with_encoder_nonlocal_attn: bool = True,
        with_decoder_nonlocal_attn: bool = True,
        use_flash_attention: bool = False,
        use_checkpointing: bool = False,
        use_convtranspose: bool = False,
        downsample_parameters=((2, 4, 1), (2, 4, 1), (2, 4, 1)),
        upsample_parameters=((2, 4, 1), (2, 4, 1), (2, 4, 1)),
    ) -> None:
        super().__init__()
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("AutoencoderKL expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("AutoencoderKL expects num_channels being same size of attention_levels")
        if isinstance(num_res_blocks, int):
            num_res_blocks = _tuple_rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same length as "
                             "`num_channels`.")
        if use_flash_attention:
            raise ValueError("torch.cuda.is_available() should be True but is False. Flash attention is only available for GPU.")
        self.use_convtranspose = bool(use_convtranspose)
        if spatial_dims not in (2, 3):
            raise ValueError("spatial_dims must be 2 or 3")

        sd = self.spatial_dims = spatial_dims
        self.groups, self.eps = norm_num_groups, norm_eps
        self.latent_channels = latent_channels
        self.use_checkpointing = use_checkpointing
        self.in_channels, self.out_channels = in_channels, out_channels
        ch, nrb, att = tuple(num_channels), tuple(num_res_blocks), tuple(attention_levels)
        L = len(ch)
        ds = [tuple(item) for item in downsample_parameters]  # (stride, kernel, padding), AEKL:703-705
        us = [tuple(item) for item in upsample_parameters]
        spec = ParamSpec(self, sd)
        self._enc, self._dec = [], []
        self._attns = []
        k3 = _axis3(3, sd, 1)
        s1 = (1, 1, 1)
        p1 = _axis3(1, sd, 0)

        def add(plan, prefix, kind, *args, sub=""):
            name = f"{prefix}.blocks.{len(plan)}{sub}"
            plan.append((kind, name) + args)
            return name

        def res(plan, prefix, cin, cout):  # ResBlock (AEKL:150-189)
            n = add(plan, prefix, "res")
            spec.norm(n + ".norm1", cin)
            spec.conv(n + ".conv1.conv", cin, cout, 3)
            spec.norm(n + ".norm2", cout)
            spec.conv(n + ".conv2.conv", cout, cout, 3)
            if cin != cout:
                spec.conv(n + ".nin_shortcut.conv", cin, cout, 1)

        def attn(plan, prefix, c):
            n = add(plan, prefix, "attn")
            spec.attention(n, c)
            self._attns.append(n)

        def conv(plan, prefix, cin, cout, stride, kernel, padding, sub):
            n = add(plan, prefix, "conv", _axis3(kernel, sd, 1), _axis3(stride, sd, 1), _axis3(padding, sd, 0), sub=sub)
            spec.conv(n, cin, cout, kernel)

        # Encoder (AEKL:369-465)
        Ep = self._enc
        conv(Ep, "encoder", in_channels, ch[0], ds[0][0], ds[0][1], ds[0][2], ".conv")
        out_c = ch[0]
        for i in range(L):
            in_c, out_c = out_c, ch[i]
            for _ in range(nrb[i]):
                res(Ep, "encoder", in_c, out_c)
                in_c = out_c
                if att[i]:
                    attn(Ep, "encoder", in_c)
            if i != L - 1:
                conv(Ep, "encoder", in_c, in_c, ds[i + 1][0], ds[i + 1][1], ds[i + 1][2], ".conv.conv")
        if with_encoder_nonlocal_attn:
            res(Ep, "encoder", ch[-1], ch[-1]); attn(Ep, "encoder", ch[-1]); res(Ep, "encoder", ch[-1], ch[-1])
        spec.norm(add(Ep, "encoder", "norm"), ch[-1])
        conv(Ep, "encoder", ch[-1], latent_channels, 1, 3, 1, ".conv")

        # Decoder (AEKL:518-617)
        Dp = self._dec
        rch, ratt, rnrb = list(reversed(ch)), list(reversed(att)), list(reversed(nrb))
        conv(Dp, "decoder", latent_channels, rch[0], 1, 3, 1, ".conv")
        if with_decoder_nonlocal_attn:
            res(Dp, "decoder", rch[0], rch[0]); attn(Dp, "decoder", rch[0]); res(Dp, "decoder", rch[0], rch[0])
        out_c = rch[0]
        for i in range(L):
            in_c, out_c = out_c, rch[i]
            for _ in range(rnrb[i]):
                res(Dp, "decoder", in_c, out_c)
                in_c = out_c
                if ratt[i]:
                    attn(Dp, "decoder", in_c)
            if i != L - 1 and use_convtranspose:  # Upsample = ConvTranspose(stride, kernel, padding) (AEKL:66-77); weight [in, out, k..]
                n = add(Dp, "decoder", "upT", _axis3(us[i][1], sd, 1), _axis3(us[i][0], sd, 1), _axis3(us[i][2], sd, 0))
                spec.conv(n + ".conv.conv", in_c, in_c, us[i][1])
            elif i != L - 1:  # Upsample: nearest x stride, then a FIXED k3/p1 conv (AEKL:78-86, 99-105)
                n = add(Dp, "decoder", "up", tuple(int(v) for v in _axis3(us[i][0], sd, 1)))
                spec.conv(n + ".conv.conv", in_c, in_c, 3)
        spec.norm(add(Dp, "decoder", "norm"), in_c)
        conv(Dp, "decoder", in_c, out_channels, 1, 3, 1, ".conv")

        for q in ("quant_conv_mu", "quant_conv_log_sigma", "post_quant_conv"):  # AEKL:723-749
            spec.conv(q + ".conv", latent_channels, latent_channels, 1)

        groups = []
        for a in self._attns:
            groups.append([f"{a}.to_{t}.weight" for t in "qkv"])
            groups.append([f"{a}.to_{t}.bias" for t in "qkv"])
        self._init_plumbing(spec, groups)
        self._k3, self._s1, self._p1 = k3, s1, p1
        # `self.encoder` is the parameter container created by ParamSpec; the reference's trainers read these two
        # attributes from it (T-LDM:267, 513, 519)
        self.encoder.spatial_dims, self.encoder.in_channels = spatial_dims, in_channels

    # ------------------------------------------------------------------------------------------ engine
    def _run_plan(self, c: E.Ctx, plan, x, need_dx):
        if self.use_checkpointing and c.tape is not None:  # h = torch.utils.checkpoint.checkpoint(self.encoder | self.decoder, x)
            return E.checkpoint(c, lambda cc, xx: self._run_plan_body(cc, plan, xx, need_dx), x)
        return self._run_plan_body(c, plan, x, need_dx)

    def _run_plan_body(self, c: E.Ctx, plan, x, need_dx):
        pending_norm = None
        first = True
        for step in plan:
            kind, name = step[0], step[1]
            if kind == "conv":
                x = E.conv(c, x, name, step[2], step[3], step[4], norm=pending_norm, silu=False, need_dx=need_dx or not first)
                pending_norm = None
            elif kind == "res":  # ResBlock.forward (AEKL:191-204)
                n1 = E.gn(c, x, name + ".norm1", self.groups, self.eps)
                h = E.conv(c, x, name + ".conv1.conv", self._k3, self._s1, self._p1, norm=n1, silu=True)
                n2 = E.gn(c, h, name + ".norm2", self.groups, self.eps)
                xs = x
                if name + ".nin_shortcut.conv.weight" in c.arena.offsets:
                    xs = E.conv(c, x, name + ".nin_shortcut.conv", (1, 1, 1), self._s1, (0, 0, 0), bias_grad_like=name + ".conv2.conv")
                x = E.conv(c, h, name + ".conv2.conv", self._k3, self._s1, self._p1, norm=n2, silu=True, res=xs)
            elif kind == "attn":
                x = E.attention(c, x, name, self.groups, self.eps, 1)  # num_head_channels=None -> one head (AEKL:235)
            elif kind == "norm":  # GroupNorm directly followed by a conv, no activation (AEKL:450-463, 604-615)
                pending_norm = E.gn(c, x, name, self.groups, self.eps)
            elif kind == "up":
                x = E.upsample_conv(c, x, name + ".conv.conv", step[2], self._k3, self._p1)
            elif kind == "upT":
                x = E.conv_transpose(c, x, name + ".conv.conv", step[2], step[3], step[4])
            first = False
        return x

    def _encode_run(self, c: E.Ctx, x_cl, need_dx):
        h = self._run_plan(c, self._enc, x_cl, need_dx)
        one, zero = (1, 1, 1), (0, 0, 0)
        z_mu = E.conv(c, h, "quant_conv_mu.conv", one, one, zero)
        lv = E.conv(c, h, "quant_conv_log_sigma.conv", one, one, zero)
        sigma = torch.empty_like(lv)
        call("mi_logvar_to_sigma_fwd", ptr(lv), ptr(sigma), lv.numel())
        if c.tape is not None:
            tape = c.tape

            def bwd():
                ds = tape.take(sigma)
                if ds is not None:
                    dlv = torch.empty_like(lv)
                    call("mi_logvar_to_sigma_bwd", ptr(ds), ptr(lv), ptr(sigma), ptr(dlv), lv.numel())
                    tape.put(lv, dlv)

            tape.record(bwd)
        return z_mu, sigma

    def _decode_run(self, c: E.Ctx, z_cl, need_dx):
        one, zero = (1, 1, 1), (0, 0, 0)
        z = E.conv(c, z_cl, "post_quant_conv.conv", one, one, zero, need_dx=need_dx)
        return self._run_plan(c, self._dec, z, True)

    # ------------------------------------------------------------------------------------------ reference API
    def _check(self, x, channels):
        if not x.is_cuda:
            raise RuntimeError("medical_image_generation_amd runs on MI355X only: move the module and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        if x.shape[1] != channels:
            raise ValueError(f"expected {channels} channels, got {x.shape[1]}")

    def _edge(self, runner, nouts, x):
        grad_enabled = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        return _NetFn.apply(self, runner, nouts, grad_enabled, x, None, *self.parameters())

    def encode(self, x):
        self._check(x, self.in_channels)

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            mu, sigma = self._encode_run(c, x_cl, need_dx)
            return (mu, sigma), {"x_cl": x_cl}

        return self._edge(runner, 2, x)

    def sampling(self, z_mu, z_sigma):
        eps = torch.randn_like(z_sigma)  # AEKL:786-787
        return z_mu + eps * z_sigma

    def decode(self, z):
        self._check(z, self.latent_channels)

        def runner(c, zin, need_dx):
            z_cl = ops.to_channels_last(zin.contiguous().float())
            return (self._decode_run(c, z_cl, need_dx),), {"x_cl": z_cl}

        return self._edge(runner, 1, z)

    def reconstruct(self, x):
        z_mu, _ = self.encode(x)
        return self.decode(z_mu)

    def forward(self, x):
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        return self.decode(z), z_mu, z_sigma

    def encode_stage_2_inputs(self, x):
        z_mu, z_sigma = self.encode(x)
        return self.sampling(z_mu, z_sigma)

    def decode_stage_2_outputs(self, z):
        return self.decode(z)
