"""DiffusionModelUNet "with strides" on the MI355X HIP path.

Drop-in for the reference class (medimgen/diffusion_model_unet_with_strides.py:1713-2021): same constructor
keywords, same `forward(x, timesteps, ...)`, same `state_dict()` names and shapes (checkpoints round-trip), same
ValueError conditions.  Forward and backward run entirely on hand-written HIP kernels (channels-last bf16
activations, fp32 accumulate / statistics / gradients); PyTorch provides memory, streams and the autograd edge.

Also on the HIP path: class embeddings (`num_class_embeds`), `resblock_updown=True` (avg-pool / nearest resnet resamplers,
UNet:640-644, 679-687), ControlNet residual inputs (UNet:1995-2010) and `with_conditioning=True` (SpatialTransformer blocks:
self-attention, cross-attention on `context`, GEGLU feed-forward, UNet:72-342; `dropout_cattn` must be 0).  Not implemented:
xformers flash attention (`use_flash_attention=True` raises like the reference does without xformers).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import nn

from . import engine as E
from . import hipops as ops
from ._lib import call, ptr

F32 = torch.float32
import os as _os  # noqa: E402
SKIP_SLOTS = _os.environ.get("MI_SKIP_SLOTS", "1") == "1"  # skips written straight into their concatenation buffers (_run); 0: copied


class _Tree(nn.Module):
    """Anonymous container: gives parameters the reference's dotted state_dict names."""


def _attach(root, dotted, tensor):
    mod = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, _Tree())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _tuple_rep(v, n):
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError(f"Sequence must have length {n}, got {len(v)}.")
        return tuple(v)
    return (v,) * n


def _axis3(v, sd, fill):
    """per-axis value for our (D, H, W) convention; 2-D nets get a unit leading axis."""
    t = _tuple_rep(v, sd)
    return (fill,) * (3 - sd) + tuple(int(a) for a in t)


class ParamSpec:
    """Declares parameters (torch default initialisers, like the nn.Conv/Linear/GroupNorm the reference builds)."""

    def __init__(self, root, sd):
        self.root, self.sd = root, sd
        self.order = []  # (name, shape, trainable) in ARENA order

    def _add(self, name, t, trainable=True):
        _attach(self.root, name, t)
        self.order.append((name, tuple(t.shape), trainable))

    def conv(self, name, cin, cout, k, zero=False):
        k = _tuple_rep(k, self.sd)
        w, b = torch.empty(cout, cin, *k), torch.empty(cout)
        if zero:  # zero_module (UNet:63-69)
            w.zero_(), b.zero_()
        else:
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            bound = 1 / math.sqrt(cin * math.prod(k))
            nn.init.uniform_(b, -bound, bound)
        self._add(name + ".weight", w)
        self._add(name + ".bias", b)

    def linear(self, name, cin, cout, trainable=True, bias=True):
        w = torch.empty(cout, cin)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin)
        self._add(name + ".weight", w, trainable)
        if bias:
            self._add(name + ".bias", torch.empty(cout).uniform_(-bound, bound), trainable)

    def transformer(self, name, c, cdim, layers):
        """SpatialTransformer parameters (UNet:256-312; inner_dim == c)."""
        name = name + "." if name else ""
        self.norm(name + "norm", c)
        self.conv(name + "proj_in.conv", c, c, 1)
        for k in range(layers):
            blk = f"{name}transformer_blocks.{k}"
            for attn, kv in (("attn1", c), ("attn2", cdim)):  # CrossAttention (UNet:86-111): q / k / v without bias
                self.linear(f"{blk}.{attn}.to_q", c, c, bias=False)
                self.linear(f"{blk}.{attn}.to_k", kv, c, bias=False)
                self.linear(f"{blk}.{attn}.to_v", kv, c, bias=False)
                self.linear(f"{blk}.{attn}.to_out.0", c, c)
            self.linear(f"{blk}.ff.linear1", c, 8 * c)  # monai MLPBlock(hidden c, mlp_dim 4c, act="GEGLU"), UNet:211
            self.linear(f"{blk}.ff.linear2", 4 * c, c)
            for n in ("norm1", "norm2", "norm3"):
                self.norm(f"{blk}.{n}", c)
        self.conv(name + "proj_out.conv", c, c, 1, zero=True)

    def norm(self, name, c):
        self._add(name + ".weight", torch.ones(c))
        self._add(name + ".bias", torch.zeros(c))

    def attention(self, name, c):
        pre = name + "." if name else ""
        self.norm(pre + "norm", c)
        for n in ("to_q", "to_k", "to_v"):
            self.linear(pre + n, c, c)
        self.linear(pre + "proj_attn", c, c, trainable=False)  # constructed, never called (UNet:383 vs 418-458)


def arena_order(entries, groups, late=()):
    """Reorder (name, shape, trainable) so that every list in `groups` is adjacent and in the given order, then move the tensors
    named in `late` (gradients complete last in the backward pass; a group is late as a whole or not at all) to the front, keeping
    the relative order: the arena becomes [late | early | untrainable] (engine.ParamArena)."""
    grouped = {n for g in groups for n in g}
    by_name = {e[0]: e for e in entries}
    out = []
    for g in groups:
        out.extend(by_name[n] for n in g)
    out.extend(e for e in entries if e[0] not in grouped)
    late = set(late)
    for g in groups:
        assert len({n in late for n in g}) <= 1, f"group straddles the late / early split: {g[0]} ..."
    return [e for e in out if e[0] in late and e[2]] + [e for e in out if not (e[0] in late and e[2])]


class HipModule(nn.Module):
    """Shared plumbing: flat parameter arena on the device, conv-plan cache, autograd edge."""

    def _init_plumbing(self, spec: ParamSpec, groups, late=()):
        self._late = tuple(n for n in late)
        self._entries = arena_order(spec.order, groups, self._late)
        self._arena = None
        self._plans = {}
        self._packb = None

    def arena(self, device) -> E.ParamArena:
        """(Re)bind every nn.Parameter to a view of one flat device buffer; idempotent while nothing moved."""
        params = dict(self.named_parameters())
        a = self._arena
        first = self._entries[0][0]
        if a is None or a.data.device != device or params[first].data_ptr() != a.view(first).data_ptr():
            a = E.ParamArena(self._entries, device, late=self._late)
            for name, _, _ in self._entries:
                v = a.view(name)
                v.copy_(params[name].data)
                params[name].data = v
            self._arena = a
            self._plans = {}
            self._packb = None
        return a

    def pack_all(self):
        """Re-pack every conv's weights in one launch; returns the plan keys it covered (empty before the plans exist).
        Plans created later (another input shape) are packed individually by engine.conv and join the batch next time."""
        if not self._plans or self._arena is None:
            return ()
        pb = getattr(self, "_packb", None)
        if pb is None or len(pb.keys) != len(self._plans):
            if torch.cuda.is_current_stream_capturing():
                return () if pb is None else pb.run()
            pb = self._packb = E.PackBatch(self._arena, self._plans)
        return pb.run()

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .float(): parameters move, the arena is rebuilt lazily
        self._arena = None
        self._packb = None
        return super()._apply(fn, *args, **kwargs)


class _NetFn(torch.autograd.Function):
    """Autograd edge of a whole network (or block): forward = tape forward, backward = tape backward on HIP kernels.
    `emb` is an optional second differentiable input (the time embedding of a stand-alone ResnetBlock); runners that take it
    return extra["d_emb"], a callable evaluated after the tape has run."""

    @staticmethod
    def forward(ctx, module, runner, nouts, grad_enabled, x, emb, *params):
        dev = x.device
        ctx.set_materialize_grads(False)  # unused outputs (e.g. z_sigma) arrive as None, not as zero tensors
        arena = module.arena(dev)
        c = E.Ctx(arena, module._plans, grad_enabled=grad_enabled, prepacked=module.pack_all())  # (autograd disables grad mode inside forward)
        need_dx = bool(grad_enabled and ctx.needs_input_grad[4])
        outs_cl, extra = runner(c, x, need_dx) if emb is None else runner(c, x, need_dx, emb)
        ctx.c, ctx.module, ctx.outs_cl, ctx.need_dx, ctx.extra = c, module, outs_cl, need_dx, extra
        ctx.need_demb = bool(emb is not None and grad_enabled and ctx.needs_input_grad[5])
        ctx.sd = module.spatial_dims
        ctx.names = [n for n, _ in module.named_parameters()]
        outs = tuple(ops.to_channels_first(o, ctx.sd) for o in outs_cl)
        return outs if nouts > 1 else outs[0]

    @staticmethod
    def backward(ctx, *douts):
        c = ctx.c
        arena = c.arena
        arena.grad.zero_()
        tape = c.tape
        for o, do in zip(ctx.outs_cl, douts):
            if do is not None:
                tape.put(o, ops.to_channels_last(do.contiguous().float()))
        for fn in reversed(tape.fns):
            fn()
        tape.fns.clear()
        E.join_side(arena.grad.device)
        dx = None
        if ctx.need_dx:
            g = tape.take(ctx.extra["x_cl"])
            dx = ops.to_channels_first(ops.dense(g), ctx.sd) if g is not None else None
        demb = ctx.extra["d_emb"]() if ctx.need_demb else None
        trainable = {n for n, _, t in ctx.module._entries if t}
        grads = tuple(arena.gview(n).clone() if n in trainable else None for n in ctx.names)
        tape.grads.clear(), tape.keep.clear()
        return (None, None, None, None, dx, demb) + grads


class DiffusionModelUNet(HipModule):
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
        # same conditions as UNet:1766-1809
        if with_conditioning is True and cross_attention_dim is None:
            raise ValueError("DiffusionModelUNet expects dimension of the cross-attention conditioning (cross_attention_dim) "
                             "when using with_conditioning.")
        if cross_attention_dim is not None and with_conditioning is False:
            raise ValueError("DiffusionModelUNet expects with_conditioning=True when specifying the cross_attention_dim.")
        if dropout_cattn > 1.0 or dropout_cattn < 0.0:
            raise ValueError("Dropout cannot be negative or >1.0!")
        if any((c % norm_num_groups) != 0 for c in num_channels):
            raise ValueError("DiffusionModelUNet expects all num_channels being multiple of norm_num_groups")
        if len(num_channels) != len(attention_levels):
            raise ValueError("DiffusionModelUNet expects num_channels being same size of attention_levels")
        if isinstance(num_head_channels, int):
            num_head_channels = _tuple_rep(num_head_channels, len(attention_levels))
        if len(num_head_channels) != len(attention_levels):
            raise ValueError("num_head_channels should have the same length as attention_levels.")
        if isinstance(num_res_blocks, int):
            num_res_blocks = _tuple_rep(num_res_blocks, len(num_channels))
        if len(num_res_blocks) != len(num_channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same length as "
                             "`num_channels`.")
        if use_flash_attention:
            raise ValueError("use_flash_attention is True but xformers is not installed.")
        if with_conditioning and dropout_cattn > 0.0:
            raise NotImplementedError("dropout_cattn > 0 is not on the HIP path (the reference's default is 0.0)")
        if with_conditioning and (cross_attention_dim % 8 or any(a and n % 8 for a, n in zip(attention_levels, num_head_channels))):
            raise ValueError("cross_attention_dim and num_head_channels must be multiples of 8 on the HIP path")
        if spatial_dims not in (2, 3):
            raise ValueError("spatial_dims must be 2 or 3")

        sd = self.spatial_dims = spatial_dims
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = ch = tuple(num_channels)
        self.num_res_blocks = nrb = tuple(num_res_blocks)
        self.attention_levels = att = tuple(attention_levels)
        self.num_head_channels = nhc = tuple(num_head_channels)
        self.with_conditioning = bool(with_conditioning)
        self.cross_attention_dim, self.transformer_num_layers = cross_attention_dim, transformer_num_layers
        self.num_class_embeds = num_class_embeds
        self.resblock_updown = bool(resblock_updown)
        # Activation checkpointing per ResnetBlock / AttentionBlock (BASELINE config 5; the reference has it in AutoencoderKL only,
        # AEKL:761-762): not a constructor keyword of the reference class, so it is an attribute -- `net.use_checkpointing = True`
        self.use_checkpointing = False
        self.groups, self.eps = norm_num_groups, norm_eps
        L = len(ch)
        self._k = [_axis3(kernel_sizes[i], sd, 1) for i in range(L)]
        self._s = [_axis3(strides[i], sd, 1) for i in range(L)]
        self._p = [_axis3(paddings[i], sd, 0) for i in range(L)]
        temb = self.temb_dim = ch[0] * 4

        spec = ParamSpec(self, sd)
        self._resnets: list[tuple[str, int, int]] = []  # (name, cin, cout) in forward order
        self._attns: list[str] = []

        def resnet(name, cin, cout):
            spec.norm(name + ".norm1", cin)
            spec.conv(name + ".conv1.conv", cin, cout, 3)
            spec.linear(name + ".time_emb_proj", temb, cout)
            spec.norm(name + ".norm2", cout)
            spec.conv(name + ".conv2.conv", cout, cout, 3, zero=True)
            if cin != cout:
                spec.conv(name + ".skip_connection.conv", cin, cout, 1)
            self._resnets.append((name, cin, cout))

        def attn(name, c):
            if with_conditioning:  # SpatialTransformer in place of AttentionBlock (UNet:1859-1860)
                spec.transformer(name, c, cross_attention_dim, transformer_num_layers)
            else:
                spec.attention(name, c)
                self._attns.append(name)

        spec.conv("conv_in.conv", in_channels, ch[0], kernel_sizes[0])
        spec.linear("time_embed.0", ch[0], temb)
        spec.linear("time_embed.2", temb, temb)
        out_c = ch[0]
        for i in range(L):
            in_c, out_c = out_c, ch[i]
            for j in range(nrb[i]):
                resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
                if att[i]:
                    attn(f"down_blocks.{i}.attentions.{j}", out_c)
            if i != L - 1:
                if resblock_updown:  # ResnetBlock(down=True): avg-pool resampler, no parameters of its own (UNet:752-765)
                    resnet(f"down_blocks.{i}.downsampler", out_c, out_c)
                else:
                    spec.conv(f"down_blocks.{i}.downsampler.op.conv", out_c, out_c, kernel_sizes[i + 1])
        resnet("middle_block.resnet_1", ch[-1], ch[-1])
        attn("middle_block.attention", ch[-1])
        resnet("middle_block.resnet_2", ch[-1], ch[-1])
        rch, rnrb, ratt = list(reversed(ch)), list(reversed(nrb)), list(reversed(att))
        out_c = rch[0]
        for i in range(L):
            prev, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, L - 1)]
            n = rnrb[i] + 1
            for j in range(n):
                skip_c = in_c if j == n - 1 else out_c
                resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out_c) + skip_c, out_c)
                if ratt[i]:
                    attn(f"up_blocks.{i}.attentions.{j}", out_c)
            if i != L - 1:
                if resblock_updown:  # ResnetBlock(up=True) (UNet:1229-1241)
                    resnet(f"up_blocks.{i}.upsampler", out_c, out_c)
                else:
                    spec.conv(f"up_blocks.{i}.upsampler.conv.conv", out_c, out_c, 3)
        spec.norm("out.0", ch[0])
        spec.conv("out.2.conv", ch[0], out_channels, 3, zero=True)
        if num_class_embeds is not None:  # nn.Embedding(num_class_embeds, time_embed_dim): N(0, 1) rows (UNet:1837-1839)
            spec._add("class_embedding.weight", torch.randn(num_class_embeds, temb))

        # arena adjacency: one GEMM for every time_emb_proj, one [3C, C] matrix per attention block
        groups = [[r[0] + ".time_emb_proj.weight" for r in self._resnets], [r[0] + ".time_emb_proj.bias" for r in self._resnets],
                  [r[0] + ".conv1.conv.bias" for r in self._resnets]]
        for a in self._attns:
            groups.append([f"{a}.to_{t}.weight" for t in "qkv"])
            groups.append([f"{a}.to_{t}.bias" for t in "qkv"])
        # Gradient-completion order (data-parallel overlap, trainer._ArenaTrainer): the backward pass ends with the finest levels of
        # the down path, conv_in and the time-embedding MLP (which also produces every time_emb_proj / conv1-bias gradient).  Those
        # tensors form the arena's late prefix; everything else (>= 95 % of the bytes: the wide, coarse levels) is final when the
        # backward reaches the cut mark _run records after down level `_cut_level`.
        self._cut_level = max(0, L // 2 - 1)
        late_pre = tuple(f"down_blocks.{i}." for i in range(self._cut_level + 1)) + ("conv_in.", "time_embed.", "class_embedding.")
        late = [n for n, _, _ in spec.order if n.startswith(late_pre) or ".time_emb_proj." in n or n.endswith(".conv1.conv.bias")]
        self._init_plumbing(spec, groups, late)
        # Channels of the tensor each skip will be concatenated BEHIND (UNet:1263: cat([h, skip])), in the order the skips are produced:
        # a skip is written straight into the last channels of its concatenation buffer (_run), so the up path copies nothing.
        rch_, rnrb_ = list(reversed(ch)), list(reversed(nrb))
        pops = []
        for i in range(L):
            for j in range(rnrb_[i] + 1):
                pops.append((rch_[i - 1] if i > 0 else rch_[0]) if j == 0 else rch_[i])
        self._skip_ca = list(reversed(pops))
        self._temb_off = {}
        off = 0
        for name, _, cout in self._resnets:
            self._temb_off[name] = off
            off += cout
        self._temb_total = off

    # ------------------------------------------------------------------------------------------ engine forward
    def _resnet(self, c, x, name, temb_all, d_temb_all, out=None, mode=None, stride=None, kernel=None):
        """ResnetBlock.forward (UNet:674-701) with this block's columns of the all-resnets time-embedding projection."""
        cout = c.p(name + ".conv1.conv.weight").shape[0]
        off = self._temb_off[name]

        def run(cc, xx):
            return E.resnet(cc, xx, name, self.spatial_dims, self.groups, self.eps, temb_all[:, off:off + cout],
                            d_temb_all[:, off:off + cout] if d_temb_all is not None else None, mode=mode, stride=stride, kernel=kernel,
                            out=out if cc is c or cc.tape is None else None)  # the recomputation writes a buffer of its own

        return E.checkpoint(c, run, x) if self.use_checkpointing else run(c, x)

    def _attention(self, c, x, name, heads):
        if self.with_conditioning:  # heads = channels // num_head_channels (UNet:976-990)
            run = lambda cc, xx: E.spatial_transformer(cc, xx, name, self._context, self.groups, self.eps, heads, self.transformer_num_layers)
        else:
            run = lambda cc, xx: E.attention(cc, xx, name, self.groups, self.eps, heads)
        return E.checkpoint(c, run, x) if self.use_checkpointing else run(c, x)

    def _heads(self, ch, nhc):
        return ch // nhc if nhc is not None else 1

    def _run(self, c: E.Ctx, x_cl, timesteps, need_dx, class_labels=None, down_res=None, mid_res=None, context=None):
        ch, L, sd = self.block_out_channels, len(self.block_out_channels), self.spatial_dims
        self._context = context  # bf16 [B * Sc, Cc] token matrix of the cross-attention context (with_conditioning) or None
        a = c.arena
        dev = x_cl.device
        grad = c.tape is not None
        # -- time embedding MLP + every time_emb_proj in one GEMM (UNet:1966-1972, 692-695)
        t0 = ops.timestep_embedding(timesteps, ch[0])
        e1, bwd1 = E.linear_f32(t0, c.p("time_embed.0.weight"), c.p("time_embed.0.bias"), c.g("time_embed.0.weight"), c.g("time_embed.0.bias"))
        s1v = ops.silu_f32(e1)
        emb, bwd2 = E.linear_f32(s1v, c.p("time_embed.2.weight"), c.p("time_embed.2.bias"), c.g("time_embed.2.weight"), c.g("time_embed.2.bias"))
        if self.num_class_embeds is not None:  # emb = emb + class_embedding(class_labels)  (UNet:1975-1980)
            call("mi_embedding_add", ptr(emb), ptr(c.p("class_embedding.weight")), ptr(class_labels), emb.shape[0], emb.shape[1])
        se = ops.silu_f32(emb)
        wn = [r[0] + ".time_emb_proj.weight" for r in self._resnets]
        bn = [r[0] + ".time_emb_proj.bias" for r in self._resnets]
        T = self._temb_total
        temb_all, bwd3 = E.linear_f32(se, a.span(wn).view(T, self.temb_dim), a.span(bn), a.span(wn, a.grad).view(T, self.temb_dim),
                                      a.span(bn, a.grad))
        ops.add_f32_(temb_all, a.span([r[0] + ".conv1.conv.bias" for r in self._resnets]))  # fold conv1 biases in
        d_temb_all = ops.zero_f32_2d_(torch.empty_like(temb_all)) if grad else None  # conv1 wgrads accumulate their dy column sums here
        if grad:
            def bwd_emb():
                E.join_side(dev)  # (weight gradients forked onto the side stream wrote into d_temb_all)
                # d_temb_all holds every conv1's per-image dy column sums: their batch sums are the conv1 bias gradients
                ops.sum_rows_f32(d_temb_all, a.span([r[0] + ".conv1.conv.bias" for r in self._resnets], a.grad), accumulate=True)
                d_se = bwd3(d_temb_all)
                d_emb = ops.silu_bwd_f32(emb, d_se)
                if self.num_class_embeds is not None:
                    call("mi_embedding_bwd", ptr(d_emb), ptr(class_labels), ptr(c.g("class_embedding.weight")), d_emb.shape[0], d_emb.shape[1])
                d_s1 = bwd2(d_emb)
                bwd1(ops.silu_bwd_f32(e1, d_s1), need_dx=False)

            c.tape.record(bwd_emb)

        k3 = (1,) * (3 - sd) + (3,) * sd
        # Skip slots.  Every tensor the down path pushes on the skip stack is later the SECOND half of a channel concatenation with a
        # known first half (self._skip_ca).  Its producer therefore writes it straight into the last channels of that concatenation's
        # buffer (a channel-slice view: every kernel takes a voxel pitch), and the first half is written there by ITS producer when
        # the up path gets to it: torch.cat costs nothing in either direction.  Producers that cannot write through a view
        # (attention; ControlNet sums) fall back to one copy.
        slots = down_res is None and SKIP_SLOTS
        skips = []  # (tensor, concat buffer or None)

        def slot(dims, cb):
            """(buffer, view of its last cb channels) for the next skip to be produced, or (None, None)."""
            if not slots:
                return None, None
            ca = self._skip_ca[len(skips)]
            buf = torch.empty((x_cl.shape[0],) + tuple(dims) + (ca + cb,), dtype=torch.bfloat16, device=dev)
            return buf, buf[..., ca:]

        def conv_dims(dims, k, s, p):
            return tuple((d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip(dims, k, s, p))

        buf, view = slot(conv_dims(x_cl.shape[1:4], self._k[0], self._s[0], self._p[0]), ch[0])
        h = E.conv(c, x_cl, "conv_in.conv", self._k[0], self._s[0], self._p[0], need_dx=need_dx, out=view)
        skips.append((h, buf))
        for i in range(L):
            for j in range(self.num_res_blocks[i]):
                att = self.attention_levels[i]
                buf, view = (None, None) if att else slot(h.shape[1:4], ch[i])
                h = self._resnet(c, h, f"down_blocks.{i}.resnets.{j}", temb_all, d_temb_all, out=view)
                if att:
                    h = self._attention(c, h, f"down_blocks.{i}.attentions.{j}", self._heads(ch[i], self.num_head_channels[i]))
                skips.append((h, buf))
            if i != L - 1:
                if self.resblock_updown:
                    h = self._resnet(c, h, f"down_blocks.{i}.downsampler", temb_all, d_temb_all, mode="down", stride=self._s[i + 1],
                                     kernel=self._k[i + 1])
                    skips.append((h, None))
                else:
                    buf, view = slot(conv_dims(h.shape[1:4], self._k[i + 1], self._s[i + 1], self._p[i + 1]), ch[i])
                    h = E.conv(c, h, f"down_blocks.{i}.downsampler.op.conv", self._k[i + 1], self._s[i + 1], self._p[i + 1], out=view)
                    skips.append((h, buf))
            if i == self._cut_level and grad:
                c.tape.record(E.CUT)  # backward order: everything recorded after this point has run when the tape gets here
        if down_res is not None:  # ControlNet residuals: one per skip, added before the up path reads them (UNet:1995-2003)
            if len(down_res) != len(skips):
                raise ValueError(f"down_block_additional_residuals must hold {len(skips)} tensors, got {len(down_res)}")
            skips = [(E.add(c, sk, r, b_needs_grad=False), None) for (sk, _), r in zip(skips, down_res)]
        h = self._resnet(c, h, "middle_block.resnet_1", temb_all, d_temb_all)
        h = self._attention(c, h, "middle_block.attention", self._heads(ch[-1], self.num_head_channels[-1]))

        # Whatever feeds a skip concatenation (UNet:1263, 1377, 1504) is written by its producing conv straight into the FIRST
        # channels of the concat buffer (the skip's slot buffer when it has one), and in backward its gradient is a view of d(cat).
        def cat_buffer(ca):
            sk, sbuf = skips[-1]
            if sbuf is not None and sbuf.shape[-1] == ca + sk.shape[-1]:
                return sbuf, sbuf[..., :ca]
            return E.concat_buffer(sk.shape, ca, sk.shape[-1], dev)

        pending, view = cat_buffer(ch[-1])
        h = self._resnet(c, h, "middle_block.resnet_2", temb_all, d_temb_all, out=None if mid_res is not None else view)
        if mid_res is not None:  # UNet:2008-2010
            h = E.add(c, h, mid_res, b_needs_grad=False)
        rch, rnrb = list(reversed(ch)), list(reversed(self.num_res_blocks))
        ratt, rnhc = list(reversed(self.attention_levels)), list(reversed(self.num_head_channels))
        rs, rp = list(reversed(self._s)), list(reversed(self._p))
        for i in range(L):
            n_res = rnrb[i] + 1
            for j in range(n_res):
                sk, sbuf = skips.pop()
                if pending is None or (sbuf is not None and pending is not sbuf):  # h was not produced into a buffer: use the skip's
                    pending = sbuf if sbuf is not None and sbuf.shape[-1] == h.shape[-1] + sk.shape[-1] else pending
                h = E.concat(c, h, sk, buf=pending)
                pending, view = (None, None)
                if j < n_res - 1 and not ratt[i]:  # this resnet's output is the next concat's first half
                    pending, view = cat_buffer(rch[i])
                h = self._resnet(c, h, f"up_blocks.{i}.resnets.{j}", temb_all, d_temb_all, out=view)
                if ratt[i]:
                    h = self._attention(c, h, f"up_blocks.{i}.attentions.{j}", self._heads(rch[i], rnhc[i]))
            if i != L - 1 and self.resblock_updown:
                pending, view = cat_buffer(rch[i])
                h = self._resnet(c, h, f"up_blocks.{i}.upsampler", temb_all, d_temb_all, out=view, mode="up", stride=rs[i])
            elif i != L - 1:  # Upsample.forward (UNet:569-588): nearest x stride, then k3 conv with the LEVEL's padding
                pending, view = cat_buffer(rch[i])
                h = E.upsample_conv(c, h, f"up_blocks.{i}.upsampler.conv.conv", rs[i], k3, rp[i], out=view)
        no = E.gn(c, h, "out.0", self.groups, self.eps)
        p1 = (0,) * (3 - sd) + (1,) * sd
        return E.conv(c, h, "out.2.conv", k3, (1, 1, 1), p1, norm=no, silu=True)

    # ------------------------------------------------------------------------------------------ public forward
    def forward(self, x, timesteps, context=None, class_labels=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None):
        if context is not None and not self.with_conditioning:
            raise ValueError("model should have with_conditioning = True if context is provided")
        if timesteps.ndim != 1:
            raise ValueError("Timesteps should be a 1d-array")
        if x.shape[1] != self.in_channels:
            raise ValueError(f"Input number of channels ({x.shape[1]}) is not equal to expected number of channels ({self.in_channels})")
        if not x.is_cuda:
            raise RuntimeError("medical_image_generation_amd runs on MI355X only: move the module and inputs to 'cuda' "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        timesteps = timesteps.to(device=x.device, dtype=torch.int64)
        if self.num_class_embeds is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            class_labels = class_labels.to(device=x.device, dtype=torch.int64).contiguous()
            if class_labels.shape != timesteps.shape or int(class_labels.min()) < 0 or int(class_labels.max()) >= self.num_class_embeds:
                raise IndexError("class_labels must hold one index in [0, num_class_embeds) per sample")  # nn.Embedding's range check

        # ControlNet residuals (UNet:1995-2010) enter as constants: a ControlNet is trained through its own graph, not through the
        # U-Net's skip additions, so no gradient is returned for them
        to_cl = lambda r: ops.to_channels_last(r.detach().to(x.device).contiguous().float())
        down_res = [to_cl(r) for r in down_block_additional_residuals] if down_block_additional_residuals is not None else None
        mid_res = to_cl(mid_block_additional_residual) if mid_block_additional_residual is not None else None

        ctx_tokens = None
        if context is not None:  # [B, Sc, cross_attention_dim] -> bf16 token matrix; a constant of the step (no gradient returned)
            if context.dim() != 3 or context.shape[0] != x.shape[0] or context.shape[2] != self.cross_attention_dim:
                raise ValueError(f"context must be [batch, tokens, {self.cross_attention_dim}], got {tuple(context.shape)}")
            ctx_tokens = ops.cast_bf16(context.detach().to(x.device).contiguous().float().reshape(-1, context.shape[2]))

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            y = self._run(c, x_cl, timesteps, need_dx, class_labels, down_res, mid_res, ctx_tokens)
            return (y,), {"x_cl": x_cl}

        grad_enabled = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        return _NetFn.apply(self, runner, 1, grad_enabled, x, None, *self.parameters())
