"""Oracle networks: functional fp32 restatement of the reference's two model files.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Written as a flat parameter tree
plus functional forwards -- deliberately not a class-per-block port -- so that
it shares no structure with the HIP-backed modules it checks.

Reference (relative to /root/reference/medimgen/):
  UNet = diffusion_model_unet_with_strides.py, AEKL = autoencoderkl_with_strides.py
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- helpers
class _Tree(nn.Module):
    """Anonymous container so parameters can live under the reference's dotted names."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Tree())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _rep(v, n: int) -> tuple:
    """monai.utils.ensure_tuple_rep as used at UNet:1786,1795 / AEKL:679."""
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError(f"sequence must have length {n}, got {len(v)}")
        return tuple(v)
    return (v,) * n


def _conv_nd(sd: int):
    return {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[sd]


class _Builder:
    """Creates parameters with torch's default initialisers (what nn.Conv/nn.Linear/
    nn.GroupNorm inside the reference do) under the reference's state_dict names."""

    def __init__(self, root: nn.Module, sd: int):
        self.root, self.sd = root, sd

    def conv(self, name: str, cin: int, cout: int, k, zero: bool = False) -> None:
        k = _rep(k, self.sd)
        w = torch.empty(cout, cin, *k)
        b = torch.empty(cout)
        if zero:  # zero_module, UNet:63-69
            w.zero_(), b.zero_()
        else:
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            bound = 1 / math.sqrt(cin * math.prod(k))
            nn.init.uniform_(b, -bound, bound)
        _attach(self.root, name + ".weight", w)
        _attach(self.root, name + ".bias", b)

    def linear(self, name: str, cin: int, cout: int) -> None:
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin)
        _attach(self.root, name + ".weight", w)
        _attach(self.root, name + ".bias", torch.empty(cout).uniform_(-bound, bound))

    def norm(self, name: str, c: int) -> None:
        _attach(self.root, name + ".weight", torch.ones(c))
        _attach(self.root, name + ".bias", torch.zeros(c))

    def linear_nobias(self, name: str, cin: int, cout: int) -> None:
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        _attach(self.root, name + ".weight", w)

    def transformer(self, name: str, c: int, cdim: int, layers: int) -> None:
        """SpatialTransformer.__init__, UNet:256-312 (inner_dim == c: heads * num_head_channels)."""
        self.norm(name + ".norm", c)
        self.conv(name + ".proj_in.conv", c, c, 1)
        for k in range(layers):
            blk = f"{name}.transformer_blocks.{k}"
            for attn, kv in (("attn1", c), ("attn2", cdim)):  # CrossAttention.__init__, UNet:86-111 (q/k/v without bias)
                self.linear_nobias(f"{blk}.{attn}.to_q", c, c)
                self.linear_nobias(f"{blk}.{attn}.to_k", kv, c)
                self.linear_nobias(f"{blk}.{attn}.to_v", kv, c)
                self.linear(f"{blk}.{attn}.to_out.0", c, c)
            self.linear(f"{blk}.ff.linear1", c, 8 * c)  # MLPBlock(hidden c, mlp_dim 4c, GEGLU): third-party monai, UNet:211
            self.linear(f"{blk}.ff.linear2", 4 * c, c)
            for n in ("norm1", "norm2", "norm3"):
                self.norm(f"{blk}.{n}", c)
        self.conv(name + ".proj_out.conv", c, c, 1, zero=True)

    def attention(self, name: str, c: int) -> None:
        # AttentionBlock.__init__, UNet:377-383 / AEKL:238-244 (proj_attn exists, is never used)
        self.norm(name + ".norm", c)
        for n in ("to_q", "to_k", "to_v", "proj_attn"):
            self.linear(f"{name}.{n}", c, c)


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """UNet:461-485."""
    if timesteps.ndim != 1:
        raise ValueError("Timesteps should be a 1d-array")
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def self_attention(p: dict, name: str, x: torch.Tensor, groups: int, eps: float, heads: int) -> torch.Tensor:
    """AttentionBlock.forward, UNet:418-458 == AEKL:283-323.  No output projection."""
    b, c = x.shape[:2]
    h = F.group_norm(x, groups, p[name + ".norm.weight"], p[name + ".norm.bias"], eps)
    h = h.reshape(b, c, -1).transpose(1, 2)  # [B, S, C]
    q = F.linear(h, p[name + ".to_q.weight"], p[name + ".to_q.bias"])
    k = F.linear(h, p[name + ".to_k.weight"], p[name + ".to_k.bias"])
    v = F.linear(h, p[name + ".to_v.weight"], p[name + ".to_v.bias"])
    s = q.shape[1]
    d = c // heads

    def split(t):
        return t.reshape(b, s, heads, d).permute(0, 2, 1, 3)  # [B, H, S, d]

    scale = 1 / math.sqrt(c / heads)
    att = torch.softmax(torch.matmul(split(q), split(k).transpose(-1, -2)) * scale, dim=-1)
    o = torch.matmul(att, split(v)).permute(0, 2, 1, 3).reshape(b, s, c)
    return o.transpose(1, 2).reshape(x.shape) + x


def cross_attention(p: dict, name: str, x: torch.Tensor, context, heads: int) -> torch.Tensor:
    """CrossAttention.forward, UNet:156-175: x [B, S, C]; context [B, Sc, Cc] or None (self-attention)."""
    ctx = x if context is None else context
    q = F.linear(x, p[name + ".to_q.weight"])
    k = F.linear(ctx, p[name + ".to_k.weight"])
    v = F.linear(ctx, p[name + ".to_v.weight"])
    b, s, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)

    att = torch.softmax(torch.matmul(split(q), split(k).transpose(-1, -2)) * (1 / math.sqrt(d)), dim=-1)
    o = torch.matmul(att, split(v)).permute(0, 2, 1, 3).reshape(b, s, c)
    return F.linear(o, p[name + ".to_out.0.weight"], p[name + ".to_out.0.bias"])


def spatial_transformer(p: dict, name: str, x: torch.Tensor, context, groups: int, eps: float, heads: int, layers: int) -> torch.Tensor:
    """SpatialTransformer.forward, UNet:314-342, with BasicTransformerBlock.forward (UNet:225-234) inlined.  The feed-forward is
    monai's MLPBlock(act="GEGLU") (third-party; restated: linear1 -> x * gelu(gate) -> linear2)."""
    sd = x.dim() - 2
    c = x.shape[1]
    h = F.group_norm(x, groups, p[name + ".norm.weight"], p[name + ".norm.bias"], eps)
    h = _conv_nd(sd)(h, p[name + ".proj_in.conv.weight"], p[name + ".proj_in.conv.bias"])
    t = h.reshape(x.shape[0], c, -1).transpose(1, 2)  # [B, S, C]
    for k in range(layers):
        blk = f"{name}.transformer_blocks.{k}"
        ln = lambda n, v: F.layer_norm(v, (c,), p[f"{blk}.{n}.weight"], p[f"{blk}.{n}.bias"], 1e-5)
        t = cross_attention(p, blk + ".attn1", ln("norm1", t), None, heads) + t
        t = cross_attention(p, blk + ".attn2", ln("norm2", t), context, heads) + t
        a, gate = F.linear(ln("norm3", t), p[blk + ".ff.linear1.weight"], p[blk + ".ff.linear1.bias"]).chunk(2, dim=-1)
        t = F.linear(a * F.gelu(gate), p[blk + ".ff.linear2.weight"], p[blk + ".ff.linear2.bias"]) + t
    h = t.transpose(1, 2).reshape(x.shape)
    return _conv_nd(sd)(h, p[name + ".proj_out.conv.weight"], p[name + ".proj_out.conv.bias"]) + x


# --------------------------------------------------------------------------- UNet
class DiffusionModelUNet(nn.Module):
    """Restates UNet:1713-2021 (constructor surface UNet:1740-1764)."""

    def __init__(
        self,
        spatial_dims: int,
        in_channels: int,
        out_channels: int,
        num_res_blocks: Sequence[int] | int = (2, 2, 2, 2),
        num_channels: Sequence[int] = (32, 64, 64, 64),
        attention_levels: Sequence[bool] = (False, False, True, True),
        norm_num_groups: int = 32,
        norm_eps: float = 1e-6,
        resblock_updown: bool = False,
        num_head_channels: int | Sequence[int] = 8,
        with_conditioning: bool = False,
        transformer_num_layers: int = 1,
        cross_attention_dim: int | None = None,
        num_class_embeds: int | None = None,
        upcast_attention: bool = False,
        use_flash_attention: bool = False,
        dropout_cattn: float = 0.0,
        strides=((2, 2, 2), (2, 2, 2), (2, 2, 2)),
        kernel_sizes=((4, 4, 4), (4, 4, 4), (4, 4, 4)),
        paddings=(1, 1, 1),
    ) -> None:
        super().__init__()
        # UNet:1766-1809
        if with_conditioning is True and cross_attention_dim is None:
            raise ValueError("cross_attention_dim is required when with_conditioning is True")
        if cross_attention_dim is not None and with_conditioning is False:
            raise ValueError("with_conditioning must be True when cross_attention_dim is given")
        if dropout_cattn > 1.0 or dropout_cattn < 0.0:
            raise ValueError("Dropout cannot be negative or >1.0!")
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("all num_channels must be multiples of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("num_channels and attention_levels must have the same length")
        if isinstance(num_head_channels, int):
            num_head_channels = _rep(num_head_channels, len(attention_levels))
        if len(num_head_channels) != len(attention_levels):
            raise ValueError("num_head_channels must have the same length as attention_levels")
        if isinstance(num_res_blocks, int):
            num_res_blocks = _rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("num_res_blocks must be an int or have the same length as num_channels")
        if use_flash_attention:
            raise ValueError("use_flash_attention needs xformers + CUDA; not available to the oracle")

        sd = self.sd = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = ch = tuple(num_channels)
        self.num_res_blocks = nrb = tuple(num_res_blocks)
        self.attention_levels = att = tuple(attention_levels)
        self.num_head_channels = nhc = tuple(num_head_channels)
        self.groups, self.eps = norm_num_groups, norm_eps
        self.resblock_updown = resblock_updown
        self.num_class_embeds = num_class_embeds
        self.with_conditioning, self.cross_attention_dim, self.transformer_num_layers = with_conditioning, cross_attention_dim, transformer_num_layers
        self.strides, self.kernel_sizes, self.paddings = strides, kernel_sizes, paddings
        L = len(ch)
        b = _Builder(self, sd)
        temb = ch[0] * 4

        def resnet(name, cin, cout):
            # ResnetBlock.__init__, UNet:628-672
            b.norm(name + ".norm1", cin)
            b.conv(name + ".conv1.conv", cin, cout, 3)
            b.linear(name + ".time_emb_proj", temb, cout)
            b.norm(name + ".norm2", cout)
            b.conv(name + ".conv2.conv", cout, cout, 3, zero=True)
            if cin != cout:
                b.conv(name + ".skip_connection.conv", cin, cout, 1)

        def attn_params(name, c):  # AttentionBlock, or SpatialTransformer when with_conditioning (UNet:1859-1860)
            if with_conditioning:
                b.transformer(name, c, cross_attention_dim, transformer_num_layers)
            else:
                b.attention(name, c)

        b.conv("conv_in.conv", in_channels, ch[0], kernel_sizes[0])
        b.linear("time_embed.0", ch[0], temb)
        b.linear("time_embed.2", temb, temb)
        if num_class_embeds is not None:
            _attach(self, "class_embedding.weight", torch.randn(num_class_embeds, temb))

        out_c = ch[0]
        for i in range(L):  # UNet:1844-1872
            in_c, out_c = out_c, ch[i]
            for j in range(nrb[i]):
                resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
                if att[i]:
                    attn_params(f"down_blocks.{i}.attentions.{j}", out_c)
            if i != L - 1:
                if resblock_updown:
                    resnet(f"down_blocks.{i}.downsampler", out_c, out_c)
                else:
                    b.conv(f"down_blocks.{i}.downsampler.op.conv", out_c, out_c, kernel_sizes[i + 1])

        resnet("middle_block.resnet_1", ch[-1], ch[-1])
        attn_params("middle_block.attention", ch[-1])
        resnet("middle_block.resnet_2", ch[-1], ch[-1])

        rch = list(reversed(ch))
        rnrb = list(reversed(nrb))
        ratt = list(reversed(att))
        out_c = rch[0]
        for i in range(L):  # UNet:1897-1928
            prev, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, L - 1)]
            n = rnrb[i] + 1
            for j in range(n):  # UNet:1213-1226
                skip_c = in_c if j == n - 1 else out_c
                resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out_c) + skip_c, out_c)
                if ratt[i]:
                    attn_params(f"up_blocks.{i}.attentions.{j}", out_c)
            if i != L - 1:
                if resblock_updown:
                    resnet(f"up_blocks.{i}.upsampler", out_c, out_c)
                else:
                    b.conv(f"up_blocks.{i}.upsampler.conv.conv", out_c, out_c, 3)
        b.norm("out.0", ch[0])
        b.conv("out.2.conv", ch[0], out_channels, 3, zero=True)

    # -- functional pieces ---------------------------------------------------------
    def _conv(self, p, name, x, stride=1, padding=0):
        return _conv_nd(self.sd)(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)

    def _resnet(self, p, name, x, emb, mode=None, stride=None, kernel=None):
        """ResnetBlock.forward, UNet:674-701 (mode = None | 'up' | 'down' for resblock_updown)."""
        h = F.silu(F.group_norm(x, self.groups, p[name + ".norm1.weight"], p[name + ".norm1.bias"], self.eps))
        if mode == "up":  # Upsample(use_conv=False), UNet:642, 580
            sf = stride if isinstance(stride, int) else tuple(stride)
            x, h = F.interpolate(x, scale_factor=sf, mode="nearest"), F.interpolate(h, scale_factor=sf, mode="nearest")
        elif mode == "down":  # Pool[AVG](kernel_size, stride), UNet:522, 644
            pool = {1: F.avg_pool1d, 2: F.avg_pool2d, 3: F.avg_pool3d}[self.sd]
            x, h = pool(x, kernel, stride), pool(h, kernel, stride)
        h = self._conv(p, name + ".conv1.conv", h, 1, 1)
        t = F.linear(F.silu(emb), p[name + ".time_emb_proj.weight"], p[name + ".time_emb_proj.bias"])
        h = h + t.reshape(*t.shape, *([1] * self.sd))
        h = F.silu(F.group_norm(h, self.groups, p[name + ".norm2.weight"], p[name + ".norm2.bias"], self.eps))
        h = self._conv(p, name + ".conv2.conv", h, 1, 1)
        if name + ".skip_connection.conv.weight" in p:
            x = self._conv(p, name + ".skip_connection.conv", x, 1, 0)
        return x + h

    def _heads(self, c, nhc):
        return c // nhc if nhc is not None else 1

    def forward(self, x, timesteps, context=None, class_labels=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None):
        p = dict(self.named_parameters())
        ch, L = self.block_out_channels, len(self.block_out_channels)
        emb = timestep_embedding(timesteps, ch[0]).to(x.dtype)
        emb = F.linear(F.silu(F.linear(emb, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                       p["time_embed.2.weight"], p["time_embed.2.bias"])
        if self.num_class_embeds is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            emb = emb + p["class_embedding.weight"][class_labels].to(x.dtype)
        if context is not None and not self.with_conditioning:
            raise ValueError("model should have with_conditioning = True if context is provided")

        def attend(name, h, c, nhc):
            if self.with_conditioning:  # SpatialTransformer(num_attention_heads = c // num_head_channels), UNet:976-990
                return spatial_transformer(p, name, h, context, self.groups, self.eps, c // nhc, self.transformer_num_layers)
            return self_attention(p, name, h, self.groups, self.eps, self._heads(c, nhc))

        h = self._conv(p, "conv_in.conv", x, self.strides[0], self.paddings[0])
        skips = [h]
        for i in range(L):
            for j in range(self.num_res_blocks[i]):
                h = self._resnet(p, f"down_blocks.{i}.resnets.{j}", h, emb)
                if self.attention_levels[i]:
                    h = attend(f"down_blocks.{i}.attentions.{j}", h, ch[i], self.num_head_channels[i])
                skips.append(h)
            if i != L - 1:
                if self.resblock_updown:
                    h = self._resnet(p, f"down_blocks.{i}.downsampler", h, emb, "down",
                                     self.strides[i + 1], self.kernel_sizes[i + 1])
                else:
                    h = self._conv(p, f"down_blocks.{i}.downsampler.op.conv", h, self.strides[i + 1], self.paddings[i + 1])
                skips.append(h)
        if down_block_additional_residuals is not None:  # UNet:1995-2003
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]

        h = self._resnet(p, "middle_block.resnet_1", h, emb)
        h = attend("middle_block.attention", h, ch[-1], self.num_head_channels[-1])
        h = self._resnet(p, "middle_block.resnet_2", h, emb)
        if mid_block_additional_residual is not None:
            h = h + mid_block_additional_residual

        rch = list(reversed(ch))
        rnrb = list(reversed(self.num_res_blocks))
        ratt = list(reversed(self.attention_levels))
        rnhc = list(reversed(self.num_head_channels))
        rstr, rpad = list(reversed(self.strides)), list(reversed(self.paddings))
        for i in range(L):
            for j in range(rnrb[i] + 1):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self._resnet(p, f"up_blocks.{i}.resnets.{j}", h, emb)
                if ratt[i]:
                    h = attend(f"up_blocks.{i}.attentions.{j}", h, rch[i], rnhc[i])
            if i != L - 1:
                if self.resblock_updown:
                    h = self._resnet(p, f"up_blocks.{i}.upsampler", h, emb, "up", rstr[i])
                else:  # Upsample.forward, UNet:569-588: nearest x stride, then k3 conv with the LEVEL's padding
                    sf = rstr[i] if isinstance(rstr[i], int) else tuple(float(s) for s in rstr[i])
                    h = F.interpolate(h, scale_factor=sf, mode="nearest")
                    h = self._conv(p, f"up_blocks.{i}.upsampler.conv.conv", h, 1, rpad[i])
        h = F.silu(F.group_norm(h, self.groups, p["out.0.weight"], p["out.0.bias"], self.eps))
        return self._conv(p, "out.2.conv", h, 1, 1)


# --------------------------------------------------------------------------- AutoencoderKL
class AutoencoderKL(nn.Module):
    """Restates AEKL:625-834 (constructor surface AEKL:648-668)."""

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
        if any((c % norm_num_groups) != 0 for c in num_channels):  # AEKL:672-690
            raise ValueError("AutoencoderKL expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("AutoencoderKL expects num_channels being same size of attention_levels")
        if isinstance(num_res_blocks, int):
            num_res_blocks = _rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("num_res_blocks must be an int or have the same length as num_channels")
        if use_flash_attention:
            raise ValueError("use_flash_attention needs xformers + CUDA; not available to the oracle")
        # use_convtranspose=True (AEKL:66-77): Upsample = monai Convolution(is_transposed=True).  monai is not under /root/reference:
        # restated from its published form -- nn.ConvTranspose{2,3}d(kernel, stride, padding, output_padding = stride - 1), weight
        # [in, out, k..] -- PARITY UNPINNED.
        self.use_convtranspose = bool(use_convtranspose)

        sd = self.sd = spatial_dims
        self.groups, self.eps = norm_num_groups, norm_eps
        self.latent_channels = latent_channels
        self.use_checkpointing = use_checkpointing
        ch, nrb, att = tuple(num_channels), tuple(num_res_blocks), tuple(attention_levels)
        L = len(ch)
        ds = [tuple(item) for item in downsample_parameters]  # (stride, kernel, padding) per entry, AEKL:703-705
        us = [tuple(item) for item in upsample_parameters]
        b = _Builder(self, sd)
        self.enc_plan: list[tuple] = []
        self.dec_plan: list[tuple] = []

        def add(plan, prefix, kind, *args, sub=""):
            """Append a plan step; `sub` is the extra nesting between blocks.N and the parameters."""
            name = f"{prefix}.blocks.{len(plan)}"
            plan.append((kind, name + sub) + args)
            return name + sub

        def res(plan, prefix, cin, cout):  # ResBlock, AEKL:150-189
            n = add(plan, prefix, "res")
            b.norm(n + ".norm1", cin)
            b.conv(n + ".conv1.conv", cin, cout, 3)
            b.norm(n + ".norm2", cout)
            b.conv(n + ".conv2.conv", cout, cout, 3)
            if cin != cout:
                b.conv(n + ".nin_shortcut.conv", cin, cout, 1)

        def attn(plan, prefix, c):
            b.attention(add(plan, prefix, "attn"), c)

        # Encoder, AEKL:369-465
        E = self.enc_plan
        b.conv(add(E, "encoder", "conv", ds[0][0], ds[0][2], sub=".conv"), in_channels, ch[0], ds[0][1])
        out_c = ch[0]
        for i in range(L):
            in_c, out_c = out_c, ch[i]
            for _ in range(nrb[i]):
                res(E, "encoder", in_c, out_c)
                in_c = out_c
                if att[i]:
                    attn(E, "encoder", in_c)
            if i != L - 1:
                # Downsample wraps a Convolution named 'conv' (AEKL:121): blocks.N.conv.conv.*
                b.conv(add(E, "encoder", "conv", ds[i + 1][0], ds[i + 1][2], sub=".conv.conv"), in_c, in_c, ds[i + 1][1])
        if with_encoder_nonlocal_attn:
            res(E, "encoder", ch[-1], ch[-1]); attn(E, "encoder", ch[-1]); res(E, "encoder", ch[-1], ch[-1])
        b.norm(add(E, "encoder", "norm"), ch[-1])
        b.conv(add(E, "encoder", "conv", 1, 1, sub=".conv"), ch[-1], latent_channels, 3)

        # Decoder, AEKL:518-617
        D = self.dec_plan
        rch, ratt, rnrb = list(reversed(ch)), list(reversed(att)), list(reversed(nrb))
        b.conv(add(D, "decoder", "conv", 1, 1, sub=".conv"), latent_channels, rch[0], 3)
        if with_decoder_nonlocal_attn:
            res(D, "decoder", rch[0], rch[0]); attn(D, "decoder", rch[0]); res(D, "decoder", rch[0], rch[0])
        out_c = rch[0]
        for i in range(L):
            in_c, out_c = out_c, rch[i]
            for _ in range(rnrb[i]):
                res(D, "decoder", in_c, out_c)
                in_c = out_c
                if ratt[i]:
                    attn(D, "decoder", in_c)
            if i != L - 1 and use_convtranspose:
                n = add(D, "decoder", "upT", us[i][0], us[i][2])  # ConvTranspose(stride, kernel, padding), AEKL:66-77
                b.conv(n + ".conv.conv", in_c, in_c, us[i][1])  # (weight [in, out, k..]: the same shape, both are in_c)
            elif i != L - 1:
                n = add(D, "decoder", "up", us[i][0])  # nearest x stride + fixed k3/p1 conv, AEKL:78-86, 99-105
                b.conv(n + ".conv.conv", in_c, in_c, 3)
        b.norm(add(D, "decoder", "norm"), in_c)
        b.conv(add(D, "decoder", "conv", 1, 1, sub=".conv"), in_c, out_channels, 3)

        for q in ("quant_conv_mu", "quant_conv_log_sigma", "post_quant_conv"):  # AEKL:723-749
            b.conv(q + ".conv", latent_channels, latent_channels, 1)

        # attributes the reference's trainers read (T-LDM:267, 513, 519)
        self.encoder.spatial_dims, self.encoder.in_channels = spatial_dims, in_channels

    def _conv(self, p, name, x, stride=1, padding=0):
        return _conv_nd(self.sd)(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)

    def _run(self, plan, x):
        p = dict(self.named_parameters())
        for step in plan:
            kind, name = step[0], step[1]
            if kind == "conv":
                x = self._conv(p, name, x, step[2], step[3])
            elif kind == "res":  # ResBlock.forward, AEKL:191-204
                h = F.silu(F.group_norm(x, self.groups, p[name + ".norm1.weight"], p[name + ".norm1.bias"], self.eps))
                h = self._conv(p, name + ".conv1.conv", h, 1, 1)
                h = F.silu(F.group_norm(h, self.groups, p[name + ".norm2.weight"], p[name + ".norm2.bias"], self.eps))
                h = self._conv(p, name + ".conv2.conv", h, 1, 1)
                if name + ".nin_shortcut.conv.weight" in p:
                    x = self._conv(p, name + ".nin_shortcut.conv", x, 1, 0)
                x = x + h
            elif kind == "attn":
                x = self_attention(p, name, x, self.groups, self.eps, 1)  # num_head_channels=None -> 1 head, AEKL:235
            elif kind == "norm":
                x = F.group_norm(x, self.groups, p[name + ".weight"], p[name + ".bias"], self.eps)
            elif kind == "upT":
                st_, pd_ = step[2], step[3]
                op_ = st_ - 1 if isinstance(st_, int) else tuple(int(v) - 1 for v in st_)
                fn = F.conv_transpose3d if self.sd == 3 else F.conv_transpose2d
                x = fn(x, p[name + ".conv.conv.weight"], p[name + ".conv.conv.bias"], stride=st_, padding=pd_, output_padding=op_)
            elif kind == "up":
                sf = step[2] if isinstance(step[2], int) else tuple(float(s) for s in step[2])
                x = F.interpolate(x, scale_factor=sf, mode="nearest")
                x = self._conv(p, name + ".conv.conv", x, 1, 1)
        return x

    def encode(self, x):  # AEKL:753-771
        p = dict(self.named_parameters())
        if self.use_checkpointing:
            h = torch.utils.checkpoint.checkpoint(lambda t: self._run(self.enc_plan, t), x, use_reentrant=False)
        else:
            h = self._run(self.enc_plan, x)
        z_mu = self._conv(p, "quant_conv_mu.conv", h)
        z_log_var = torch.clamp(self._conv(p, "quant_conv_log_sigma.conv", h), -30.0, 20.0)
        return z_mu, torch.exp(z_log_var / 2)

    def sampling(self, z_mu, z_sigma, eps=None):  # AEKL:773-788 (eps injectable for tests)
        if eps is None:
            eps = torch.randn_like(z_sigma)
        return z_mu + eps * z_sigma

    def decode(self, z):  # AEKL:804-819
        p = dict(self.named_parameters())
        z = self._conv(p, "post_quant_conv.conv", z)
        if self.use_checkpointing:
            return torch.utils.checkpoint.checkpoint(lambda t: self._run(self.dec_plan, t), z, use_reentrant=False)
        return self._run(self.dec_plan, z)

    def reconstruct(self, x):
        return self.decode(self.encode(x)[0])

    def forward(self, x, eps=None):  # AEKL:821-825
        z_mu, z_sigma = self.encode(x)
        return self.decode(self.sampling(z_mu, z_sigma, eps)), z_mu, z_sigma

    def encode_stage_2_inputs(self, x, eps=None):
        z_mu, z_sigma = self.encode(x)
        return self.sampling(z_mu, z_sigma, eps)

    def decode_stage_2_outputs(self, z):
        return self.decode(z)
