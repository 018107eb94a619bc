"""The building blocks of the two "with strides" networks as stand-alone drop-in modules on the HIP path.

Same constructor signatures, `forward` arguments, `state_dict()` names and ValueErrors as the reference classes
(medimgen/diffusion_model_unet_with_strides.py: `ResnetBlock` :591-701, `AttentionBlock` :345-458, `Downsample` :488-531,
`Upsample` :534-588; medimgen/autoencoderkl_with_strides.py: `ResBlock` :136-204).  Each block is one autograd edge over the
tape engine -- the same engine ops the whole networks run, so a block-level parity test exercises exactly the network's kernels.
fp32 NC[D]HW in / out, channels-last bf16 inside.
"""
from __future__ import annotations

import torch

from . import engine as E
from . import hipops as ops
from .unet import HipModule, ParamSpec, _axis3, _NetFn


class _Block(HipModule):
    def _check(self, x, channels, msg):
        if x.shape[1] != channels:
            raise ValueError(msg)
        if not x.is_cuda:
            raise RuntimeError("medical_image_generation_amd runs on MI355X only: move the module and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")

    def _edge(self, runner, x, emb=None):
        grad_enabled = torch.is_grad_enabled() and (x.requires_grad or (emb is not None and emb.requires_grad)
                                                    or any(p.requires_grad for p in self.parameters()))
        return _NetFn.apply(self, runner, 1, grad_enabled, x, emb, *self.parameters())


class ResnetBlock(_Block):
    """Residual block with timestep conditioning (UNet:591-701)."""

    def __init__(self, spatial_dims: int, in_channels: int, temb_channels: int, out_channels: int | None = None, up: bool = False,
                 down: bool = False, norm_num_groups: int = 32, norm_eps: float = 1e-6, kernel_size=2, stride=4, padding=1) -> None:
        super().__init__()
        sd = self.spatial_dims = spatial_dims
        self.channels, self.emb_channels = in_channels, temb_channels
        self.out_channels = out_channels or in_channels
        self.up, self.down = up, down
        self.groups, self.eps = norm_num_groups, norm_eps
        self._mode = "up" if up else ("down" if down else None)
        self._stride, self._kernel = _axis3(stride, sd, 1), _axis3(kernel_size, sd, 1)
        spec = ParamSpec(self, sd)
        spec.norm("norm1", in_channels)
        spec.conv("conv1.conv", in_channels, self.out_channels, 3)
        spec.linear("time_emb_proj", temb_channels, self.out_channels)
        spec.norm("norm2", self.out_channels)
        spec.conv("conv2.conv", self.out_channels, self.out_channels, 3, zero=True)
        if self.out_channels != in_channels:
            spec.conv("skip_connection.conv", in_channels, self.out_channels, 1)
        self._init_plumbing(spec, [])

    def forward(self, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
        self._check(x, self.channels, f"expected {self.channels} input channels, got {x.shape[1]}")

        def runner(c, xin, need_dx, emb_in):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            e = emb_in.detach().contiguous().float()
            se = ops.silu_f32(e)  # temb = time_emb_proj(nonlinearity(emb)) (UNet:692-695)
            temb, bwd = E.linear_f32(se, c.p("time_emb_proj.weight"), c.p("time_emb_proj.bias"), c.g("time_emb_proj.weight"),
                                     c.g("time_emb_proj.bias"))
            ops.add_f32_(temb, c.p("conv1.conv.bias"))
            d_temb = ops.zero_f32_2d_(torch.empty_like(temb)) if c.tape is not None else None
            state = {}
            if c.tape is not None:
                def bwd_emb():  # runs last: every conv1 column sum is in d_temb by then
                    ops.sum_rows_f32(d_temb, c.g("conv1.conv.bias"), accumulate=True)
                    state["d_emb"] = ops.silu_bwd_f32(e, bwd(d_temb))

                c.tape.record(bwd_emb)
            # a leading identity op so that the block input has a gradient slot even when it is only read by norm1 / the resamplers
            y = E.resnet(c, x_cl, "", self.spatial_dims, self.groups, self.eps, temb, d_temb, mode=self._mode, stride=self._stride,
                         kernel=self._kernel)
            return (y,), {"x_cl": x_cl, "d_emb": lambda: state.get("d_emb")}

        return self._edge(runner, x, emb)


class AttentionBlock(_Block):
    """Self-attention over spatial positions with q / k / v Linear layers and NO output projection: `proj_attn` is constructed and
    never called (UNet:383 vs 418-458), so it holds parameters that never receive a gradient."""

    def __init__(self, spatial_dims: int, num_channels: int, num_head_channels: int | None = None, norm_num_groups: int = 32,
                 norm_eps: float = 1e-6, use_flash_attention: bool = False) -> None:
        super().__init__()
        if use_flash_attention:
            raise ValueError("use_flash_attention is True but xformers is not installed.")
        self.spatial_dims, self.num_channels = spatial_dims, num_channels
        self.num_heads = num_channels // num_head_channels if num_head_channels is not None else 1
        self.groups, self.eps = norm_num_groups, norm_eps
        spec = ParamSpec(self, spatial_dims)
        spec.attention("", num_channels)
        self._init_plumbing(spec, [[f"to_{t}.weight" for t in "qkv"], [f"to_{t}.bias" for t in "qkv"]])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check(x, self.num_channels, f"expected {self.num_channels} input channels, got {x.shape[1]}")

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            return (E.attention(c, x_cl, "", self.groups, self.eps, self.num_heads),), {"x_cl": x_cl}

        return self._edge(runner, x)


class Downsample(_Block):
    """Strided convolution (use_conv=True) or average pooling (UNet:488-531)."""

    def __init__(self, spatial_dims: int, num_channels: int, use_conv: bool, out_channels: int | None = None, stride=2, kernel_size=4,
                 padding=1) -> None:
        super().__init__()
        sd = self.spatial_dims = spatial_dims
        self.num_channels, self.out_channels, self.use_conv = num_channels, out_channels or num_channels, use_conv
        self._k, self._s, self._p = _axis3(kernel_size, sd, 1), _axis3(stride, sd, 1), _axis3(padding, sd, 0)
        spec = ParamSpec(self, sd)
        if use_conv:
            spec.conv("op.conv", num_channels, self.out_channels, kernel_size)
        elif self.num_channels != self.out_channels:
            raise ValueError("num_channels and out_channels must be equal when use_conv=False")
        self._init_plumbing(spec, [])

    def forward(self, x: torch.Tensor, emb: torch.Tensor | None = None) -> torch.Tensor:
        del emb
        self._check(x, self.num_channels, f"Input number of channels ({x.shape[1]}) is not equal to expected number of channels "
                                          f"({self.num_channels})")

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            if self.use_conv:
                y = E.conv(c, x_cl, "op.conv", self._k, self._s, self._p, need_dx=need_dx)
            else:
                y = E.avg_pool(c, x_cl, self._k, self._s)
            return (y,), {"x_cl": x_cl}

        return self._edge(runner, x)


class Upsample(_Block):
    """Nearest-neighbour up-sampling by `stride`, then an optional k3 s1 conv with `padding` (UNet:534-588)."""

    def __init__(self, spatial_dims: int, num_channels: int, use_conv: bool, out_channels: int | None = None, stride=2, padding=1) -> None:
        super().__init__()
        sd = self.spatial_dims = spatial_dims
        self.num_channels, self.out_channels, self.use_conv, self.stride = num_channels, out_channels or num_channels, use_conv, stride
        self._s, self._p = tuple(int(v) for v in _axis3(stride, sd, 1)), _axis3(padding, sd, 0)
        spec = ParamSpec(self, sd)
        if use_conv:
            spec.conv("conv.conv", num_channels, self.out_channels, 3)
        self._init_plumbing(spec, [])

    def forward(self, x: torch.Tensor, emb: torch.Tensor | None = None) -> torch.Tensor:
        del emb
        self._check(x, self.num_channels, "Input channels should be equal to num_channels")
        sd = self.spatial_dims
        k3 = (1,) * (3 - sd) + (3,) * sd

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            if self.use_conv:
                y = E.upsample_conv(c, x_cl, "conv.conv", self._s, k3, self._p)
            else:
                y = E.upsample(c, x_cl, self._s)
            return (y,), {"x_cl": x_cl}

        return self._edge(runner, x)


class ResBlock(_Block):
    """AutoencoderKL residual block: GN-SiLU-conv twice + (1x1 conv) shortcut (AEKL:136-204)."""

    def __init__(self, spatial_dims: int, in_channels: int, norm_num_groups: int, norm_eps: float, out_channels: int) -> None:
        super().__init__()
        sd = self.spatial_dims = spatial_dims
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.groups, self.eps = norm_num_groups, norm_eps
        spec = ParamSpec(self, sd)
        spec.norm("norm1", in_channels)
        spec.conv("conv1.conv", in_channels, self.out_channels, 3)
        spec.norm("norm2", self.out_channels)
        spec.conv("conv2.conv", self.out_channels, self.out_channels, 3)
        if self.in_channels != self.out_channels:
            spec.conv("nin_shortcut.conv", in_channels, self.out_channels, 1)
        self._init_plumbing(spec, [])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check(x, self.in_channels, f"expected {self.in_channels} input channels, got {x.shape[1]}")
        sd = self.spatial_dims
        k3, s1, p1 = (1,) * (3 - sd) + (3,) * sd, (1, 1, 1), (0,) * (3 - sd) + (1,) * sd

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            n1 = E.gn(c, x_cl, "norm1", self.groups, self.eps)
            h = E.conv(c, x_cl, "conv1.conv", k3, s1, p1, norm=n1, silu=True)
            n2 = E.gn(c, h, "norm2", self.groups, self.eps)
            xs = x_cl
            if "nin_shortcut.conv.weight" in c.arena.offsets:
                xs = E.conv(c, x_cl, "nin_shortcut.conv", (1, 1, 1), s1, (0, 0, 0), bias_grad_like="conv2.conv")
            return (E.conv(c, h, "conv2.conv", k3, s1, p1, norm=n2, silu=True, res=xs),), {"x_cl": x_cl}

        return self._edge(runner, x)


class SpatialTransformer(_Block):
    """Transformer block for image-like data (UNet:237-342): GroupNorm -> 1x1 proj_in -> num_layers x [self-attention,
    cross-attention on `context`, GEGLU feed-forward] -> 1x1 proj_out (zero-initialised) + input.  The feed-forward is monai's
    MLPBlock(act="GEGLU") (third-party; restated).  `context`: [B, tokens, cross_attention_dim] or None (attn2 then attends to the
    block's own tokens, UNet:159); it is treated as a constant (no gradient is returned for it)."""

    def __init__(self, spatial_dims: int, in_channels: int, num_attention_heads: int, num_head_channels: int, num_layers: int = 1,
                 dropout: float = 0.0, norm_num_groups: int = 32, norm_eps: float = 1e-6, cross_attention_dim: int | None = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False) -> None:
        super().__init__()
        if use_flash_attention:
            raise ValueError("use_flash_attention is True but xformers is not installed.")
        if dropout > 0.0:
            raise NotImplementedError("dropout > 0 is not on the HIP path (the reference's default is 0.0)")
        if num_attention_heads * num_head_channels != in_channels:
            raise NotImplementedError("the HIP path covers inner_dim == in_channels (how every block of the U-Net constructs it, UNet:976-990)")
        self.spatial_dims, self.in_channels = spatial_dims, in_channels
        self.heads, self.layers, self.groups, self.eps = num_attention_heads, num_layers, norm_num_groups, norm_eps
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else in_channels
        spec = ParamSpec(self, spatial_dims)
        spec.transformer("", in_channels, self.cross_attention_dim, num_layers)
        self._init_plumbing(spec, [])

    def forward(self, x: torch.Tensor, context: torch.Tensor | None = None) -> torch.Tensor:
        self._check(x, self.in_channels, f"expected {self.in_channels} input channels, got {x.shape[1]}")
        ctx_tokens = None
        if context is not None:
            if context.dim() != 3 or context.shape[0] != x.shape[0] or context.shape[2] != self.cross_attention_dim:
                raise ValueError(f"context must be [batch, tokens, {self.cross_attention_dim}], got {tuple(context.shape)}")
            ctx_tokens = ops.cast_bf16(context.detach().to(x.device).contiguous().float().reshape(-1, context.shape[2]))

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            return (E.spatial_transformer(c, x_cl, "", ctx_tokens, self.groups, self.eps, self.heads, self.layers),), {"x_cl": x_cl}

        return self._edge(runner, x)
