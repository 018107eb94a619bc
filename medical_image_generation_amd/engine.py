"""Tape engine: forward ops on channels-last bf16 activations that record their own HIP backward.

Why not one torch.autograd.Function per kernel: the fused kernels need to know about each other
(GroupNorm statistics feed the NEXT conv's prologue; the conv's data gradient feeds the GroupNorm/SiLU
backward which also folds in the other gradient branches of the same tensor), and gradient accumulation
must run through our kernels, not aten::add.  The tape gives that control; `functional.py` wraps whole
blocks / networks of it as torch.autograd.Function for callers that want autograd.

Everything here launches HIP kernels through hipops; torch only allocates (torch.empty / views).
"""
from __future__ import annotations

import math
import weakref

import torch

from . import _lib
from . import hipops as ops
from ._lib import call, ptr

BF16, F32 = torch.bfloat16, torch.float32

# GroupNorm(+SiLU) in front of a conv: either fused into the conv's staging (no activated tensor in HBM, but the
# transcendental is evaluated on 2.3x halo-inflated data in forward AND in wgrad) or one standalone apply pass whose bf16
# output both kernels read.  Measured on MI355X (tools/bench_conv.py): fused costs +85 us per 32-channel 128^3 conv per
# use, the standalone pass 54 us once -> standalone is the default; MI_FUSE_PROLOGUE=1 selects the fused path.
import os as _os
FUSE_PROLOGUE = _os.environ.get("MI_FUSE_PROLOGUE", "0") == "1"
# Fused (flash-style) attention for head dims 32 / 64; MI_FLASH_ATTENTION=0 forces the materialised GEMM + softmax path.
FLASH_ATTENTION = _os.environ.get("MI_FLASH_ATTENTION", "1") == "1"
# GroupNorm statistics from the per-channel sums the producing conv's epilogue emits (no statistics pass over the tensor)
FUSE_GN_STATS = ops.FUSE_GN_STATS
# A/B knob, default OFF: GroupNorm backward in two launches (partial sums by fp64 atomics into a per-pass zeroed buffer, coefficients
# derived inside the apply pass: mi_gn_bwd_fused) instead of three.  It removes the 51 finalize launches of a C4 step (~9 us each) and
# is SLOWER: 22.46 -> 23.47 ms/step (round 3, profiles/r03k_ab_gn_fused.log, same box, interleaved) -- 1024 blocks adding to the same
# 64 addresses serialise at the memory side for about as long as the finalize launch took, and every apply thread now starts with a
# dependent L2 round trip for its coefficients.
GN_BWD_FUSED = _os.environ.get("MI_GN_BWD_FUSED", "0") == "1"
# ... or only for tensors of at most this many voxels per image (the coarse levels: few blocks, so few atomics per address; there the
# finish launch it saves is a large part of the three-launch backward).  0 = off.
GN_BWD_FUSED_MAXV = int(_os.environ.get("MI_GN_BWD_FUSED_MAXV", "0"))


# A/B knob, default OFF: weight gradients of the layers whose kernels cannot fill the chip on a second stream.  A conv's data gradient
# and weight gradient only share their inputs; at the coarsest level of a batch-1 step (16^3 voxels x 256 channels: 16 tiles x 8
# channel blocks) each is a launch of 128 single-tile workgroups on 256 CUs, so the two could run side by side (captured into the
# hipGraph as two branches).  Measured (round 3, profiles/r03f_ab_side.log, same box, interleaved): 22.70 / 22.80 ms/step on one stream,
# 23.35 / 23.36 with the fork for plans of <= 160 workgroups, 23.35 / 23.42 for <= 300: the fork / join edges of 14 layers cost more
# than the overlap returns (round 2 measured the same for a fork of every layer).
SIDE_WGRAD = _os.environ.get("MI_SIDE_WGRAD", "0") == "1"
# MI_SIDE_WGRAD=2: the OTHER overlap -- chip-filling layers only.  A conv's weight gradient (MFMA-bound, one persistent workgroup of 168
# registers x 8 waves per CU: room for other waves beside it) is forked BEHIND its data gradient, so that it runs beside the GroupNorm
# backward of the conv's input (HBM-bound streaming kernels without LDS) instead of beside the data gradient (which needs the same LDS).
SIDE_BESIDE_NORM = _os.environ.get("MI_SIDE_WGRAD", "0") == "2"
SIDE_MIN_WGS = int(_os.environ.get("MI_SIDE_MIN_WGS", "256"))
DGRAD_FIRST = _os.environ.get("MI_DGRAD_FIRST", "0") == "1"  # data gradient before the weight gradient of a conv (default: after)
SIDE_MAX_WGS = int(_os.environ.get("MI_SIDE_MAX_WGS", "160"))
# A/B knob: the same weight gradients DEFERRED instead -- queued while the backward walks the coarse levels, then launched together on
# the side stream with ONE fork when the first layer with a chip-filling grid comes up (and one join at the cut mark / the end of the
# backward): the latency-bound launches of the 16^3 level (each ~40 us for ~10 us of work) run beside the finer levels' kernels.
DEFER_WGRAD = _os.environ.get("MI_DEFER_WGRAD", "0") == "1"
DEFER_MAX_TILES = int(_os.environ.get("MI_DEFER_MAX_TILES", "64"))  # layers of at most this many 4 x 8 x 8 output tiles are queued
_side_streams: dict = {}
_side_pending: set = set()  # devices whose side stream has work the main stream has not waited for
_deferred: dict = {}        # device index -> [(closure, tensors it reads)] in backward order


def _dev_index(device):
    device = torch.device(device)
    return torch.cuda.current_device() if device.index is None else device.index


def _side_stream(device):
    i = _dev_index(device)
    st = _side_streams.get(i)
    if st is None:
        st = _side_streams[i] = torch.cuda.Stream(device=i)
    return st


def flush_deferred(device):
    """Launch the queued weight gradients on the side stream, behind everything the current stream holds so far."""
    i = _dev_index(device)
    q = _deferred.get(i)
    if not q:
        return
    side = _side_stream(i)
    side.wait_stream(torch.cuda.current_stream(i))
    with torch.cuda.stream(side):
        for fn, _ in q:
            fn()
    for _, tensors in q:
        for t in tensors:
            t.record_stream(side)  # (freed blocks must not be handed out again under the side kernels)
    q.clear()
    _side_pending.add(i)


def join_side(device):
    """The current stream waits for everything forked onto the side stream (before anything reads a weight / bias gradient)."""
    i = _dev_index(device)
    flush_deferred(i)
    if i in _side_pending:
        torch.cuda.current_stream(i).wait_stream(_side_stream(i))
        _side_pending.discard(i)


def _tiles(plan):
    od, oh, ow = plan.out_dims if not isinstance(plan, ops.UpConvPlan) else plan.dims
    return plan.n * ((od + 3) // 4) * ((oh + 7) // 8) * ((ow + 7) // 8)


def _small_grid(plan):
    return plan.out_dims[0] > 1 and _tiles(plan) * ((max(plan.cin, plan.cout) + 31) // 32) <= SIDE_MAX_WGS


def _coarse(plan):
    return plan.out_dims[0] > 1 and _tiles(plan) <= DEFER_MAX_TILES


# --------------------------------------------------------------------------------------------- parameters
class ParamArena:
    """All parameters of a network in ONE flat fp32 buffer (+ one for gradients), in an order chosen so that tensors
    the kernels want fused are adjacent (q/k/v weights -> one [3C, C] matrix; every time_emb_proj -> one GEMM).
    The trainable prefix [0, n_trainable) is what the fused optimizer / DDP all-reduce touch; statically unused
    tensors (`proj_attn.*`, never called: UNet:383 vs 418-458) sit behind it and are never updated -- the same
    outcome as torch.optim skipping parameters whose grad is None."""

    def __init__(self, entries, device, late=()):
        # entries: list of (name, shape, trainable); late: names of the trainable tensors whose gradients complete LAST in the
        # backward pass (they must form a prefix of `entries`): [0, n_late) is that prefix, [n_late, n_trainable) the early segment
        # whose data-parallel all-reduce may start at the backward's cut mark (trainer._ArenaTrainer)
        self.offsets, self.shapes = {}, {}
        late = set(late)
        off = 0
        self.n_late = 0
        for trainable_pass in (True, False):
            for name, shape, trainable in entries:
                if trainable != trainable_pass:
                    continue
                n = math.prod(shape)
                self.offsets[name], self.shapes[name] = off, tuple(shape)
                off += n
                if not trainable_pass or n % 4:
                    off = (off + 3) // 4 * 4
                if trainable_pass and name in late:
                    assert self.n_late == self.offsets[name], f"late tensors must form a prefix of the arena ({name})"
                    self.n_late = off
            if trainable_pass:
                off = (off + 3) // 4 * 4
                self.n_trainable = off
        self.numel = max(off, 4)
        self.data = torch.zeros(self.numel, dtype=F32, device=device)
        self.grad = torch.zeros(self.numel, dtype=F32, device=device)

    def view(self, name, flat=None):
        flat = self.data if flat is None else flat
        o, s = self.offsets[name], self.shapes[name]
        return flat[o:o + math.prod(s)].view(s)

    def gview(self, name):
        return self.view(name, self.grad)

    def span(self, names, flat=None):
        """Contiguous view covering consecutive tensors `names` (asserts adjacency)."""
        flat = self.data if flat is None else flat
        o = self.offsets[names[0]]
        end = o
        for nme in names:
            assert self.offsets[nme] == end, f"{nme} is not adjacent in the arena"
            end += math.prod(self.shapes[nme])
        return flat[o:end]


# --------------------------------------------------------------------------------------------- tape
def CUT():
    """Tape marker (a no-op when called): recorded by a model at the forward position behind which -- in backward order: before
    which -- every gradient of the arena's early segment is final.  trainer._ArenaTrainer splits the backward there."""


class Tape:
    def __init__(self):
        self.fns = []
        self.grads = {}
        self.keep = []  # tensors whose id() is used as a key must stay alive

    def record(self, fn):
        self.fns.append(fn)

    # A tensor with two consumers (a ResnetBlock input: norm1 path + residual; a skip: next layer + its slice of d(concat)) collects
    # its gradient branches here.  Two branches stay a PAIR: the GroupNorm backward that consumes them (take2) sums them in fp32 inside
    # its own pass (mi_gn_bwd add / add2) -- the separate add kernel of a skip at 128^3 cost 77 us plus 65-79 us to densify the concat
    # slice first.  Anyone else (take) gets the sum; a third branch folds the pair.
    def take(self, t):
        g = self.grads.pop(id(t), None)
        return ops.add(g[0], g[1]) if isinstance(g, tuple) else g

    def take2(self, t):
        """(branch, second branch or None) -- for consumers that can add two streams themselves."""
        g = self.grads.pop(id(t), None)
        return g if isinstance(g, tuple) else (g, None)

    def put(self, t, g):
        """grad(t) += g."""
        cur = self.grads.get(id(t))
        if cur is None:
            self.grads[id(t)] = g
        elif isinstance(cur, tuple):
            self.grads[id(t)] = ops.add(ops.add(cur[0], cur[1]), g)
        else:
            self.grads[id(t)] = (cur, g)
        self.keep.append(t)

    def move(self, t, other: "Tape"):
        """Hand t's pending gradient (a pair stays a pair) to another tape."""
        g = self.grads.pop(id(t), None)
        if g is not None:
            other.grads[id(t)] = g
            other.keep.append(t)

    def backward(self, out, dout):
        self.put(out, dout)
        for fn in reversed(self.fns):
            fn()
        self.fns.clear()
        join_side(out.device)


class Ctx:
    """One forward/backward of one network: parameter views, plan cache, tape."""

    def __init__(self, arena: ParamArena, plans: dict, grad_enabled=True, prepacked=()):
        self.arena, self.plans = arena, plans
        self.tape = Tape() if grad_enabled else None
        self.packed = set(prepacked)  # plan keys whose weights are already packed for this pass (PackBatch.run)
        # algorithmic matmul-class flops of this pass (2*MACs; torch.utils.flop_counter convention)
        self.flops_fwd = 0
        self.flops_bwd = 0
        # id(tensor) -> (ChannelSums, weakref(tensor)): per-channel sums the producing conv emitted, so that the GroupNorm consuming
        # the tensor skips its statistics pass; id(concatenation) -> (a, b, weakref(cat)) lets a GroupNorm over a concat use both
        # halves' sums.  The keys are ids of tensors these dicts do NOT keep alive (a no-grad pass frees activations as it goes and
        # CPython reuses ids), so every lookup checks that the weak reference still IS the tensor asked about.
        self.sums = {}
        self.cat_parts = {}
        self._z64 = None   # zeroed fp64 scratch of this pass (zeros64): cleared by ONE fill, handed out in slices
        self._z64_off = 0

    def zeros64(self, numel, device, voxels=None):
        """A zeroed fp64 slice for a kernel that accumulates with atomics (the fused GroupNorm backward): carved from a buffer that one
        fill per pass clears, so that no norm needs a zero-fill node of its own.  None when the fused form is off for this size."""
        if not (GN_BWD_FUSED or (voxels is not None and voxels <= GN_BWD_FUSED_MAXV)):
            return None
        numel = (numel * ops.GN_FUSED_REPLICAS + 31) // 32 * 32  # (the kernel spreads its atomics over up to that many records)
        if self._z64 is None or self._z64_off + numel > self._z64.numel():
            self._z64 = torch.zeros(max(1 << 20, numel), dtype=torch.float64, device=device)  # 8 MiB: the ~50 norms of a C4 pass take 4
            self._z64_off = 0
        out = self._z64[self._z64_off:self._z64_off + numel]
        self._z64_off += numel
        return out

    def sums_of(self, x):
        ent = self.sums.get(id(x))
        return ent[0] if ent is not None and ent[1]() is x else None

    def parts_of(self, x):
        ent = self.cat_parts.get(id(x))
        return ent[:2] if ent is not None and ent[2]() is x else None

    def count(self, fwd_flops, dgrad=True, wgrad=True):
        self.flops_fwd += fwd_flops
        if self.tape is not None:
            self.flops_bwd += fwd_flops * (int(dgrad) + int(wgrad))

    def p(self, name):
        return self.arena.view(name)

    def g(self, name):
        return self.arena.gview(name)


class PackBatch:
    """Every conv plan of a network re-packed by ONE kernel launch (the weights change every optimizer step; 67 per-conv
    launches are mostly launch floor).  Built once all plans exist, i.e. after a first pass."""

    def __init__(self, arena: ParamArena, plans: dict):
        import ctypes as C
        self.keys = tuple(plans)
        n = len(self.keys)
        plan_arr = (C.c_void_p * n)(*[plans[k].handle for k in self.keys])
        w_arr = (C.c_void_p * n)(*[arena.view(k[0] + ".weight").data_ptr() for k in self.keys])
        h = C.c_void_p()
        _lib.call_raw("mi_conv_pack_batch_create", C.byref(h), plan_arr, w_arr, n)
        self.handle = h

    def run(self):
        call("mi_conv_pack_batch_run", self.handle)
        return self.keys

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.call_raw("mi_conv_pack_batch_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


# --------------------------------------------------------------------------------------------- ops
def gn(ctx: Ctx, x, name, groups, eps):
    """Statistics only; the affine(+SiLU) is applied by the consumer (conv prologue or gn_apply)."""
    gamma, beta = ctx.p(name + ".weight"), ctx.p(name + ".bias")
    st = None
    if FUSE_GN_STATS:
        n, v = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        ent, parts = ctx.sums_of(x), ctx.parts_of(x)
        if ent is not None:
            st = ops.gn_stats_from_sums(ent, None, n, v, groups, eps, gamma, beta)
        elif parts is not None:
            ea, eb = ctx.sums_of(parts[0]), ctx.sums_of(parts[1])
            if ea is not None and eb is not None:
                st = ops.gn_stats_from_sums(ea, eb, n, v, groups, eps, gamma, beta)
    if st is None:
        st = ops.gn_stats(x, groups, eps, gamma, beta)
    st.name = name
    return st


# Upsample (nearest x2 on all three axes) + k3 p1 conv as ONE op on the phase kernels (csrc/convph.hip: 8 taps per output voxel on
# the coarse tensor instead of 27 on an up-sampled copy); MI_UPCONV=0: upsample kernel, then the plain conv (A/B runs).
UPCONV = _os.environ.get("MI_UPCONV", "1") == "1"


def upsample_conv(ctx: Ctx, x, name, factors, kernel, padding, out=None):
    """Upsample.forward (UNet:569-588, AEKL Upsample): F.interpolate(x, scale_factor=factors, mode="nearest") then the stride-1 conv
    `name` (kernel, padding).  Fused when the layer is the 3-D factor-2 / k3 / padding-1 case every planner-generated config has."""
    n, d, h, w, cin = x.shape
    fused = UPCONV and tuple(factors) == (2, 2, 2) and tuple(kernel) == (3, 3, 3) and tuple(padding) == (1, 1, 1) and d > 1 and \
        cin % 8 == 0 and ctx.p(name + ".weight").shape[0] % 8 == 0 and x.is_contiguous()
    if not fused:
        return conv(ctx, upsample(ctx, x, factors), name, kernel, (1, 1, 1), padding, out=out)
    return conv(ctx, x, name, kernel, (1, 1, 1), padding, out=out, upconv=True)


def conv_transpose(ctx: Ctx, x, name, kernel, stride, padding):
    """AE Upsample(use_convtranspose=True) (AEKL:66-77): monai's Convolution(is_transposed=True) = nn.ConvTranspose3d(kernel, stride,
    padding, output_padding = stride - 1), weight `name.weight` [Cin, Cout, k, k, k].  A transposed convolution is the data gradient of
    the convolution C: z -> u with the same weight tensor read as [Cout_C = Cin, Cin_C = Cout]: forward = C's dgrad (the phase kernels
    of a k3 s2 p1 conv write the twice finer grid directly), input gradient = C's forward, weight gradient = C's wgrad with (dy, x) in
    the roles of (input, output gradient).  The bias rides in one affine pass over the output.  Supported: 3-D, kernel 3, padding 1,
    stride 2 (or 1) on all axes -- what per-level (stride, kernel, padding) triples of compute_downsample_parameters (CFG:751-797) give."""
    n, d, h, w, cin = x.shape
    wt = ctx.p(name + ".weight")  # [Cin, Cout, 3, 3, 3]
    cout = wt.shape[1]
    k, s_, p_ = tuple(kernel), tuple(stride), tuple(padding)
    if d <= 1 or k != (3, 3, 3) or p_ != (1, 1, 1) or s_ not in ((1, 1, 1), (2, 2, 2)) or cin % 8 or cout % 8:
        raise NotImplementedError(f"use_convtranspose: kernel {k} stride {s_} padding {p_} ({cin}->{cout}) is not on the HIP path "
                                  "(3-D, kernel 3, padding 1, stride 1 or 2 on all axes, channels in multiples of 8)")
    fine = tuple(v * s_[0] for v in (d, h, w))  # output_padding = stride - 1: exactly stride x the input extent
    key = (name, n, d, h, w)
    plan = ctx.plans.get(key)
    if plan is None:
        plan = ctx.plans[key] = ops.ConvPlan(n, fine, cout, cin, k, s_, p_)  # the convolution C on the fine grid: Cout -> Cin channels
    if key not in ctx.packed:
        plan.pack(wt)
        ctx.packed.add(key)
    y0 = plan.dgrad(x)
    bias = ctx.p(name + ".bias")
    ss = torch.stack([torch.ones_like(bias), bias], dim=1).unsqueeze(0).expand(n, cout, 2).contiguous()
    y = ops.gn_apply(y0, ops.GNStats(ss, None, 1), False)  # y = 1 * y0 + bias
    del y0
    ctx.count(2 * y.numel() * cin * 27 // (s_[0] ** 3), dgrad=True)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            plan.wgrad(dy, x, ctx.g(name + ".weight"))
            ops.colsum(dy.contiguous(), out=ctx.g(name + ".bias"), accumulate=True, merge_batch=True)
            tape.put(x, plan.fwd(dy))

        tape.record(bwd)
    return y


def conv(ctx: Ctx, x, name, kernel, stride, padding, norm=None, silu=False, addvec=None, res=None, d_addvec=None,
         need_dx=True, bias_grad_like=None, out=None, upconv=False):
    """y = conv(act(x)) + addvec + res   (weight `name.weight`, bias folded into addvec by the caller or taken from
    `name.bias` when addvec is None).  norm: GNStats of x for the fused prologue.  d_addvec: fp32 [N, Cout] view that
    receives the per-sample column sums of dy (time-embedding gradient) in backward.  bias_grad_like: name of a conv whose
    output gradient is THIS conv's output gradient (a shortcut conv added as `res` of that conv): its bias gradient, already
    computed when this backward runs, is copied instead of reducing dy a second time."""
    n, d, h, w, cin = x.shape
    wt = ctx.p(name + ".weight")
    cout = wt.shape[0]
    key = (name, n, d, h, w)
    plan = ctx.plans.get(key)
    if plan is None:
        plan = ctx.plans[key] = ops.UpConvPlan(n, (d, h, w), cin, cout) if upconv else ops.ConvPlan(n, (d, h, w), cin, cout, kernel, stride, padding)
    if key not in ctx.packed:
        plan.pack(wt)  # [Cout, Cin, (kd,) kh, kw] contiguous: same memory layout for 2-D and 3-D nets
        ctx.packed.add(key)
    av = addvec if addvec is not None else ctx.p(name + ".bias")
    if norm is not None and not FUSE_PROLOGUE:
        xin, pn, ps = ops.gn_apply(x, norm, silu), None, False
    else:
        xin, pn, ps = x, norm, silu
    y, sums = plan.fwd(xin, pn, ps, addvec=av, res=res, out=out, want_sums=True)  # out: channel-slice view of the consumer's concat buffer
    if sums is not None:
        ctx.sums[id(y)] = (sums, weakref.ref(y))
    ctx.count(2 * y.numel() * cin * math.prod(kernel), dgrad=need_dx)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            gw = ctx.g(name + ".weight")
            # column sums of dy come out of the wgrad kernel (one extra MFMA per k-step on the dY fragments it holds anyway):
            # per image into `d_addvec` (time-embedding gradient; the caller folds the rows into the bias gradient), or --
            # row pitch 0 -- summed over the batch straight into the bias gradient
            def wgrad():
                if bias_grad_like is not None:
                    plan.wgrad(xin, dy, gw, pn, ps)
                    ops.add_f32_(ctx.g(name + ".bias"), ctx.g(bias_grad_like + ".bias"))
                else:
                    plan.wgrad(xin, dy, gw, pn, ps, colsum=d_addvec if d_addvec is not None else ctx.g(name + ".bias"))

            beside_norm = (SIDE_BESIDE_NORM and need_dx and norm is not None and plan.out_dims[0] > 1 and
                           _tiles(plan) * ((max(plan.cin, plan.cout) + 31) // 32) >= SIDE_MIN_WGS)
            g_early = plan.dgrad(dy) if ((DGRAD_FIRST or beside_norm) and need_dx) else None  # (A/B: which of the two readers of dy runs first)
            di = _dev_index(dy.device)
            if SIDE_BESIDE_NORM and not beside_norm and bias_grad_like is not None:
                join_side(dy.device)  # (the bias gradient this layer copies may still be in flight on the side stream)
            if DEFER_WGRAD and need_dx and (_coarse(plan) or (bias_grad_like is not None and _deferred.get(di))):
                # (a shortcut conv copies the bias gradient its block's conv2 produces: it queues up behind a queued conv2)
                _deferred.setdefault(di, []).append((wgrad, (xin, dy)))
            elif DEFER_WGRAD and _deferred.get(di):
                flush_deferred(di)
                wgrad()
            elif beside_norm or (SIDE_WGRAD and need_dx and bias_grad_like is None and _small_grid(plan)):
                dev, side = dy.device, _side_stream(dy.device)
                side.wait_stream(torch.cuda.current_stream(dev))  # dy (and everything before it) is ready
                with torch.cuda.stream(side):
                    wgrad()
                xin.record_stream(side), dy.record_stream(side)    # (freed blocks must not be handed out again under the side kernel)
                _side_pending.add(_dev_index(dev))
            else:
                wgrad()
            if res is not None:
                tape.put(res, dy)
            if need_dx:
                g = g_early if g_early is not None else plan.dgrad(dy)
                if norm is not None:
                    other, other2 = tape.take2(x)
                    dx = ops.gn_bwd(g, x, norm, ctx.p(norm.name + ".weight"), silu, ctx.g(norm.name + ".weight"),
                                    ctx.g(norm.name + ".bias"), add=other, add2=other2, sums=ctx.zeros64(2 * x.shape[0] * x.shape[-1], x.device, x.shape[1] * x.shape[2] * x.shape[3]))
                    tape.grads[id(x)] = dx
                    tape.keep.append(x)
                else:
                    tape.put(x, g)

        tape.record(bwd)
    return y


def upsample(ctx: Ctx, x, factors):
    y = ops.upsample_nearest(x, factors)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                tape.put(x, ops.upsample_nearest_bwd(ops.dense(dy), factors))  # dy may be a slice view of d(concat)

        tape.record(bwd)
    return y


def checkpoint(ctx: Ctx, fn, x):
    """Activation checkpointing at tape level (the torch.utils.checkpoint of autoencoderkl_with_strides.py:761-762, 815-816, also
    offered per block for the U-Net): y = fn(ctx', x) runs WITHOUT a tape, so every intermediate of the segment is freed as soon
    as its consumer has run; the recorded backward re-runs fn with a tape and back-propagates through it.  The recomputation
    launches the same kernels on the same inputs, and the segment input's pending gradient is handed to the inner tape so that
    the fused "+ other branch" of the GroupNorm backward sees what it would have seen: gradients are bit-identical to the
    stored-activation path (tests/test_checkpoint_gpu.py).  Recomputed flops are not counted (SURVEY 8d convention)."""
    if ctx.tape is None:
        return fn(ctx, x)
    sub = Ctx(ctx.arena, ctx.plans, grad_enabled=False)
    sub.packed, sub.sums, sub.cat_parts = ctx.packed, ctx.sums, ctx.cat_parts  # same packed weights, same emitted channel sums
    y = fn(sub, x)
    ctx.flops_fwd += sub.flops_fwd
    ctx.flops_bwd += 2 * sub.flops_fwd
    tape = ctx.tape

    def bwd():
        dy = tape.take(y)
        if dy is None:
            return
        inner = Ctx(ctx.arena, ctx.plans, grad_enabled=True)
        inner.packed, inner.sums, inner.cat_parts = ctx.packed, ctx.sums, ctx.cat_parts
        y2 = fn(inner, x)
        it = inner.tape
        tape.move(x, it)  # (a pending pair stays a pair: the inner GroupNorm backward must see what the stored-activation path sees)
        it.backward(y2, dy)
        g = it.take(x)
        if g is not None:
            tape.grads[id(x)] = g
            tape.keep.append(x)
        it.grads.clear(), it.keep.clear()

    tape.record(bwd)
    return y


def avg_pool(ctx: Ctx, x, kernel, stride):
    """Pool[AVG](kernel_size, stride) of Downsample(use_conv=False) (UNet:522)."""
    y = ops.avg_pool(x, kernel, stride)
    if ctx.tape is not None:
        tape = ctx.tape
        dims = tuple(x.shape[1:4])

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                tape.put(x, ops.avg_pool_bwd(ops.dense(dy), dims, kernel, stride))

        tape.record(bwd)
    return y


def gn_act(ctx: Ctx, x, st, silu):
    """act(GroupNorm(x)) as a tensor of its own (for consumers other than a conv prologue: the resamplers of
    ResnetBlock(up / down), UNet:679-687).  Backward folds the other gradient branches of x in, like conv's fused path."""
    y = ops.gn_apply(x, st, silu)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            g = tape.take(y)
            if g is None:
                return
            other, other2 = tape.take2(x)
            dx = ops.gn_bwd(g, x, st, ctx.p(st.name + ".weight"), silu, ctx.g(st.name + ".weight"), ctx.g(st.name + ".bias"), add=other,
                            add2=other2, sums=ctx.zeros64(2 * x.shape[0] * x.shape[-1], x.device, x.shape[1] * x.shape[2] * x.shape[3]))
            tape.grads[id(x)] = dx
            tape.keep.append(x)

        tape.record(bwd)
    return y


def add(ctx: Ctx, a, b, b_needs_grad=True):
    """a + b (ControlNet residual inputs, UNet:1995-2010)."""
    y = ops.add(a, b)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                tape.put(a, dy)
                if b_needs_grad:
                    tape.put(b, dy)

        tape.record(bwd)
    return y


def resnet(ctx: Ctx, x, name, sd, groups, eps, addvec, d_addvec, mode=None, stride=None, kernel=None, out=None):
    """ResnetBlock.forward (UNet:674-701): 2 statistics passes (mostly from the producing convs' epilogues) + 3 fused convs.
    addvec: fp32 [N, Cout] = time_emb_proj(silu(emb)) + conv1.bias; d_addvec receives conv1's per-image dy column sums.
    mode: None | 'up' (nearest x stride) | 'down' (avg-pool kernel/stride): both x and act(norm1(x)) are resampled (UNet:679-687).
    out: where conv2 writes the block's result (the first channels of the next skip-concat buffer)."""
    k3, s1, p1 = (1,) * (3 - sd) + (3,) * sd, (1, 1, 1), (0,) * (3 - sd) + (1,) * sd
    name = name + "." if name else ""  # stand-alone blocks (blocks.ResnetBlock) have un-prefixed parameter names
    n1 = gn(ctx, x, name + "norm1", groups, eps)
    if mode is None:
        h = conv(ctx, x, name + "conv1.conv", k3, s1, p1, norm=n1, silu=True, addvec=addvec, d_addvec=d_addvec)
    else:
        h = gn_act(ctx, x, n1, True)
        if mode == "up":
            x, h = upsample(ctx, x, stride), upsample(ctx, h, stride)
        else:
            x, h = avg_pool(ctx, x, kernel, stride), avg_pool(ctx, h, kernel, stride)
        h = conv(ctx, h, name + "conv1.conv", k3, s1, p1, addvec=addvec, d_addvec=d_addvec)
    n2 = gn(ctx, h, name + "norm2", groups, eps)
    if name + "skip_connection.conv.weight" in ctx.arena.offsets:
        xs = conv(ctx, x, name + "skip_connection.conv", (1, 1, 1), s1, (0, 0, 0), bias_grad_like=name + "conv2.conv")
    else:
        xs = x
    return conv(ctx, h, name + "conv2.conv", k3, s1, p1, norm=n2, silu=True, res=xs, out=out)


def concat_buffer(a_shape, ca, cb, device):
    """Allocate the [.., ca + cb] buffer of an upcoming `concat(a, b)` and return (buffer, view of its first ca channels): the
    producer of `a` writes there directly (conv `out=`), so that half of the concatenation costs nothing."""
    buf = torch.empty(tuple(a_shape[:-1]) + (ca + cb,), dtype=BF16, device=device)
    return buf, buf[..., :ca]


def concat(ctx: Ctx, a, b, buf=None):
    """torch.cat([a, b], channel axis) (UNet:1263, 1377, 1504).  buf: the concatenation's buffer when one or both halves already
    live in it -- `a` written there by its producing conv (concat_buffer), `b` (the skip) written there when it was produced
    (unet._run: skip slots).  Only the halves that are elsewhere are copied; in backward the gradient of an in-place half is a
    channel-slice VIEW of d(cat), not a copy."""
    ca, cb = a.shape[-1], b.shape[-1]
    a_in = buf is not None and a.data_ptr() == buf.data_ptr() and ops._cs(a) == ca + cb
    b_in = buf is not None and b.data_ptr() == buf.data_ptr() + 2 * ca and ops._cs(b) == ca + cb
    if a_in or b_in:
        n, v = a.shape[0], a.shape[1] * a.shape[2] * a.shape[3]
        if not a_in:
            call("mi_copy_channels", ptr(a), ops._cs(a), 0, ptr(buf), ca + cb, 0, ca, n * v)
        if not b_in:
            call("mi_copy_channels", ptr(b), ops._cs(b), 0, ptr(buf), ca + cb, ca, cb, n * v)
        y = buf
    else:
        y = ops.concat_channels(a, b)
    ctx.cat_parts[id(y)] = (a, b, weakref.ref(y))
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            for t, lo, n_c, in_place in ((a, 0, ca, a_in), (b, ca, cb, b_in)):
                if in_place and id(t) not in tape.grads:  # first gradient of this half: hand the slice over as a strided view
                    tape.grads[id(t)] = dy[..., lo:lo + n_c]
                    tape.keep.append(t)
                    tape.keep.append(dy)
                else:
                    tape.put(t, ops.slice_channels(dy, lo, n_c))

        tape.record(bwd)
    return y


def _gemm(a, lda, sa1, sa2, b, ldb, sb1, sb2, c, ldc, sc1, sc2, m, n, k, z, z2, alpha=1.0, bias=None, res=None, ldr=0, sr1=0,
          sr2=0, accumulate=False):
    """Raw pointer-level NT GEMM: a/b/c/res are (tensor, element_offset) pairs or tensors."""

    def addr(t):
        if t is None:
            return None
        if isinstance(t, tuple):
            return t[0].data_ptr() + t[1] * t[0].element_size()
        return t.data_ptr()

    ct = c[0] if isinstance(c, tuple) else c
    call("mi_gemm_nt_bf16", addr(a), lda, sa1, sa2, addr(b), ldb, sb1, sb2, addr(c), ldc, sc1, sc2, ptr(bias), addr(res), ldr, sr1, sr2,
         m, n, k, z, z2, float(alpha), int(ct.dtype == F32), int(accumulate))


def _transpose(src, ld_in, si1, si2, rows, cols, z, z2, device):
    """[z][rows][cols] (strided) -> contiguous [z][cols][rows]."""
    out = torch.empty((z, cols, rows), dtype=BF16, device=device)
    a = src[0].data_ptr() + src[1] * 2 if isinstance(src, tuple) else src.data_ptr()
    call("mi_transpose_bf16", a, ld_in, si1, si2, ptr(out), rows, z2 * cols * rows, cols * rows, rows, cols, z, z2)
    return out


def _attention_flash(ctx, x, name, st, xn, wqkv, qkv, b, s, c, heads, scale):
    """Head dims 32 / 64: fused attention kernels (csrc/attention.hip); the projections stay on the NT GEMM."""
    dev = x.device
    d_, h_, w_ = x.shape[1:4]
    y = torch.empty_like(x)
    lse = torch.empty((b * heads, s), dtype=F32, device=dev)
    nws = _lib.call_raw("mi_attn_workspace_bytes", c, heads, b, s)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)  # partial results of the split kernels; dead after each call
    call("mi_attn_fwd", ptr(qkv), 3 * c, c, heads, b, s, float(scale), ptr(x), ptr(y), ptr(lse), ptr(ws), nws)
    del ws
    ctx.count(2 * b * s * c * 3 * c)
    ctx.count(4 * b * s * s * c)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            dqkv = torch.empty((b * s, 3 * c), dtype=BF16, device=dev)
            dsum = torch.empty((b * heads, s), dtype=F32, device=dev)
            ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
            call("mi_attn_bwd", ptr(qkv), 3 * c, c, heads, b, s, float(scale), ptr(y), ptr(x), ptr(dy), ptr(lse), ptr(dsum), ptr(dqkv),
                 ptr(ws), nws)
            _attention_param_and_input_grads(ctx, tape, x, name, st, xn, wqkv, dqkv, dy, b, s, c)

        tape.record(bwd)
    return y


def _attention_param_and_input_grads(ctx, tape, x, name, st, xn, wqkv, dqkv, dy, b, s, c):
    """Shared tail of the attention backward: projection weight/bias gradients, dx through the projections and the norm."""
    dev = x.device
    pre = name + "." if name else ""
    gw = ctx.arena.span([f"{pre}to_{t}.weight" for t in "qkv"], ctx.arena.grad)
    gb = ctx.arena.span([f"{pre}to_{t}.bias" for t in "qkv"], ctx.arena.grad)
    # dW[3C, C] += dqkv^T xn and db += colsum(dqkv) in ONE pass over the two activations (no transposes, no long-K GEMM on 12 workgroups)
    call("mi_linear_wgrad_bf16", ptr(xn), c, c, ptr(dqkv), 3 * c, 3 * c, b * s, ptr(gw), ptr(gb))
    wqkv_t = _transpose(wqkv, c, 0, 0, 3 * c, c, 1, 1, dev)[0]           # [C, 3C]
    dxn = torch.empty(x.shape, dtype=BF16, device=dev)
    _gemm(dqkv, 3 * c, 0, 0, wqkv_t, 3 * c, 0, 0, dxn, c, 0, 0, b * s, c, 3 * c, 1, 1)
    other = tape.take(x)
    dx = ops.gn_bwd(dxn, x, st, ctx.p(pre + "norm.weight"), False, ctx.g(pre + "norm.weight"), ctx.g(pre + "norm.bias"), add=dy, add2=other,
                    sums=ctx.zeros64(2 * x.shape[0] * x.shape[-1], x.device, x.shape[1] * x.shape[2] * x.shape[3]))
    tape.grads[id(x)] = dx
    tape.keep.append(x)


def attention(ctx: Ctx, x, name, groups, eps, heads):
    """AttentionBlock.forward (UNet:418-458 / AEKL:283-323): GN -> q,k,v -> softmax(QK^T/sqrt(d)) V -> + x.  No out-proj.
    Channels-last makes the reference's [B,C,S]->[B,S,C] transpose free."""
    b, d_, h_, w_, c = x.shape
    s = d_ * h_ * w_
    hd = c // heads
    scale = 1.0 / math.sqrt(c / heads)
    dev = x.device
    pre = name + "." if name else ""  # stand-alone blocks (blocks.AttentionBlock) have un-prefixed parameter names
    st = gn(ctx, x, pre + "norm", groups, eps)
    xn = ops.gn_apply(x, st, False)                                              # [B, S, C]
    wqkv = ops.cast_bf16(ctx.arena.span([f"{pre}to_{t}.weight" for t in "qkv"]).view(3 * c, c))
    bqkv = ctx.arena.span([f"{pre}to_{t}.bias" for t in "qkv"])
    qkv = torch.empty((b * s, 3 * c), dtype=BF16, device=dev)
    _gemm(xn, c, 0, 0, wqkv, c, 0, 0, qkv, 3 * c, 0, 0, b * s, 3 * c, c, 1, 1, bias=bqkv)
    z = b * heads
    sq = s * 3 * c  # batch stride of qkv
    if FLASH_ATTENTION and _lib.call_raw("mi_attn_supported", c, heads):
        return _attention_flash(ctx, x, name, st, xn, wqkv, qkv, b, s, c, heads, scale)
    # Materialised path (head dims the fused kernels do not cover).  Every bf16 matrix whose LAST axis is the token axis is a GEMM
    # operand reduced over that axis: the NT GEMM loads 16-byte pieces along K, so those matrices get a row pitch `sp` = S rounded
    # up to a multiple of 8 with zero pad columns (written by the softmax / transpose kernels themselves) and the products run
    # over K = sp.  (3^3 = 27 or 5^3 = 125 tokens at the coarsest level of non-power-of-two patches.)
    sp = (s + 7) // 8 * 8
    scores = torch.empty((z, s, s), dtype=F32, device=dev)
    _gemm((qkv, 0), 3 * c, sq, hd, (qkv, c), 3 * c, sq, hd, scores, s, heads * s * s, s * s, s, s, hd, z, heads, alpha=scale)
    probs = ops.softmax_fwd(scores, pad8=True)                                    # [z, S, S] view, pitch sp
    del scores

    def tr_tokens(src, ld_in, si1, si2, cols):
        """[z][S][cols] (strided) -> [z][cols][sp] with the token axis last, zero padded."""
        out = torch.empty((z, cols, sp), dtype=BF16, device=dev)
        a = src[0].data_ptr() + src[1] * 2 if isinstance(src, tuple) else src.data_ptr()
        call("mi_transpose_bf16", a, ld_in, si1, si2, ptr(out), sp, heads * cols * sp, cols * sp, s, cols, z, heads)
        return out

    vt = tr_tokens((qkv, 2 * c), 3 * c, sq, hd, hd)                               # [z, hd, sp]
    y = torch.empty_like(x)
    _gemm(probs, sp, heads * s * sp, s * sp, vt, sp, heads * hd * sp, hd * sp, (y, 0), c, s * c, hd, s, hd, sp, z, heads,
          res=(x, 0), ldr=c, sr1=s * c, sr2=hd)
    ctx.count(2 * b * s * c * 3 * c)       # q, k, v projections
    ctx.count(4 * b * s * s * c)           # QK^T and PV
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            dp = torch.empty((z, s, s), dtype=F32, device=dev)
            _gemm((dy, 0), c, s * c, hd, (qkv, 2 * c), 3 * c, sq, hd, dp, s, heads * s * s, s * s, s, s, hd, z, heads)
            ds = ops.softmax_bwd(probs, dp, scale)                                # [z, S, S] view, pitch sp
            del dp
            dqkv = torch.empty((b * s, 3 * c), dtype=BF16, device=dev)
            kt = tr_tokens((qkv, c), 3 * c, sq, hd, hd)
            _gemm(ds, sp, heads * s * sp, s * sp, kt, sp, heads * hd * sp, hd * sp, (dqkv, 0), 3 * c, sq, hd, s, hd, sp, z, heads)
            qt = tr_tokens((qkv, 0), 3 * c, sq, hd, hd)
            dst = tr_tokens(ds, sp, heads * s * sp, s * sp, s)                    # dS^T [z, S, sp]
            _gemm(dst, sp, heads * s * sp, s * sp, qt, sp, heads * hd * sp, hd * sp, (dqkv, c), 3 * c, sq, hd, s, hd, sp, z, heads)
            del dst, ds
            pt = tr_tokens(probs, sp, heads * s * sp, s * sp, s)                  # P^T [z, S, sp]
            dot = tr_tokens((dy, 0), c, s * c, hd, hd)
            _gemm(pt, sp, heads * s * sp, s * sp, dot, sp, heads * hd * sp, hd * sp, (dqkv, 2 * c), 3 * c, sq, hd, s, hd, sp, z, heads)
            del pt
            _attention_param_and_input_grads(ctx, tape, x, name, st, xn, wqkv, dqkv, dy, b, s, c)

        tape.record(bwd)
    return y


def linear_f32(x_f32, w, b, gw, gb):
    """Tiny fp32-in / fp32-out Linear on the embedding vectors (bf16 MFMA inside).  w [out, in], b [out] and their
    gradient buffers gw, gb are fp32 arena views.  Returns (y, backward(dy) -> dx)."""
    out_f, in_f = w.shape
    xb, wb = ops.cast_bf16(x_f32), ops.cast_bf16(w)
    m = x_f32.shape[0]
    y = torch.empty((m, out_f), dtype=F32, device=x_f32.device)
    _gemm(xb, in_f, 0, 0, wb, in_f, 0, 0, y, out_f, 0, 0, m, out_f, in_f, 1, 1, bias=b)

    def bwd(dy_f32, need_dx=True):
        """dy may be a column slice; accumulates into the weight/bias gradients, returns dx (fp32) or None."""
        dyb = ops.cast_bf16(dy_f32)
        ops.sum_rows_f32(dy_f32, gb, accumulate=True)
        mp = (m + 7) // 8 * 8  # K of the weight-gradient GEMM must be a multiple of 8: the transpose kernel zero-fills [m, mp)
        dy_t = torch.empty((out_f, mp), dtype=BF16, device=dy_f32.device)
        x_t = torch.empty((in_f, mp), dtype=BF16, device=dy_f32.device)
        call("mi_transpose_bf16", ptr(dyb), out_f, 0, 0, ptr(dy_t), mp, 0, 0, m, out_f, 1, 1)
        call("mi_transpose_bf16", ptr(xb), in_f, 0, 0, ptr(x_t), mp, 0, 0, m, in_f, 1, 1)
        _gemm(dy_t, mp, 0, 0, x_t, mp, 0, 0, gw, in_f, 0, 0, out_f, in_f, mp, 1, 1, accumulate=True)
        if not need_dx:
            return None
        w_t = _transpose(wb, in_f, 0, 0, out_f, in_f, 1, 1, dy_f32.device)[0]   # [in, out]
        dx = torch.empty((m, in_f), dtype=F32, device=dy_f32.device)
        _gemm(dyb, out_f, 0, 0, w_t, out_f, 0, 0, dx, in_f, 0, 0, m, in_f, out_f, 1, 1)
        return dx

    return y, bwd


# --------------------------------------------------------------------------------------------- cross-attention path (UNet:72-342)
def _reshape(ctx: Ctx, x, shape):
    """x viewed with another shape (same storage): channels-last activations ARE the [B*S, C] token matrix (UNet:327-330 for free)."""
    y = x.view(shape)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            g = tape.take(y)
            if g is not None:
                tape.put(x, ops.dense(g).view(x.shape))

        tape.record(bwd)
    return y


def layernorm(ctx: Ctx, x, name, eps=1e-5):
    """nn.LayerNorm(C) over token rows x [M, C] bf16 (UNet:219-221)."""
    m, c = x.shape
    gamma, beta = ctx.p(name + ".weight"), ctx.p(name + ".bias")
    y = torch.empty_like(x)
    mr = torch.empty((m, 2), dtype=F32, device=x.device)
    call("mi_layernorm_fwd", ptr(x), c, ptr(gamma), ptr(beta), ptr(y), c, ptr(mr), m, c, float(eps))
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            dx = torch.empty_like(x)
            call("mi_layernorm_bwd", ptr(dy), c, ptr(x), c, ptr(gamma), ptr(mr), ptr(dx), c, ptr(ctx.g(name + ".weight")),
                 ptr(ctx.g(name + ".bias")), m, c)
            tape.put(x, dx)

        tape.record(bwd)
    return y


def linear(ctx: Ctx, x, name, bias=True, res=None, need_dx=True):
    """nn.Linear on token rows: y = x W^T (+ b) (+ res).  x [M, K] bf16 dense, W fp32 [N, K] in the arena (bf16 MFMA, fp32 accumulate)."""
    m, k = x.shape
    w = ctx.p(name + ".weight")
    n = w.shape[0]
    if k % 8 or n % 8:
        raise ValueError(f"{name}: feature counts must be multiples of 8 on the HIP path (got {k} -> {n})")
    wb = ops.cast_bf16(w)
    y = torch.empty((m, n), dtype=BF16, device=x.device)
    _gemm(x, k, 0, 0, wb, k, 0, 0, y, n, 0, 0, m, n, k, 1, 1, bias=ctx.p(name + ".bias") if bias else None, res=res, ldr=n)
    ctx.count(2 * m * n * k, dgrad=need_dx)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            call("mi_linear_wgrad_bf16", ptr(x), k, k, ptr(dy), n, n, m, ptr(ctx.g(name + ".weight")), ptr(ctx.g(name + ".bias")) if bias else None)
            if res is not None:
                tape.put(res, dy)
            if need_dx:
                wt = _transpose(wb, k, 0, 0, n, k, 1, 1, x.device)[0]  # [K, N]
                dx = torch.empty((m, k), dtype=BF16, device=x.device)
                _gemm(dy, n, 0, 0, wt, n, 0, 0, dx, k, 0, 0, m, k, n, 1, 1)
                tape.put(x, dx)

        tape.record(bwd)
    return y


def geglu(ctx: Ctx, h):
    """GEGLU of monai's MLPBlock(act="GEGLU") (UNet:211): h [M, 2F] -> h[:, :F] * gelu(h[:, F:])."""
    m, f2 = h.shape
    f = f2 // 2
    y = torch.empty((m, f), dtype=BF16, device=h.device)
    call("mi_geglu_fwd", ptr(h), ptr(y), m, f)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                dh = torch.empty_like(h)
                call("mi_geglu_bwd", ptr(h), ptr(dy), ptr(dh), m, f)
                tape.put(h, dh)

        tape.record(bwd)
    return y


def mha(ctx: Ctx, q, k, v, b, sq, skv, heads):
    """CrossAttention._attention (UNet:135-154): softmax(Q K^T / sqrt(d)) V per head; q [B*Sq, C], k / v [B*Skv, C] bf16 dense, the
    heads are column slices (reshape_heads_to_batch_dim is free in this layout).  Materialised scores (fp32) -- the query and
    key/value token counts differ for a context -- with every token axis that becomes a GEMM reduction axis zero-padded to a
    multiple of 8 by the softmax / transpose kernels."""
    c = q.shape[1]
    hd = c // heads
    if hd % 8:
        raise ValueError("head width must be a multiple of 8 on the HIP path")
    scale = 1.0 / math.sqrt(hd)
    z, dev = b * heads, q.device
    kp, qp = (skv + 7) // 8 * 8, (sq + 7) // 8 * 8

    def tr(src, rows, pad):
        """the head slices of a [B * rows, C] token matrix -> [z][hd][pad]: token axis last, zero padded to `pad`."""
        out = torch.empty((z, hd, pad), dtype=BF16, device=dev)
        call("mi_transpose_bf16", ptr(src), c, rows * c, hd, ptr(out), pad, heads * hd * pad, hd * pad, rows, hd, z, heads)
        return out

    scores = torch.empty((z, sq, skv), dtype=F32, device=dev)
    _gemm(q, c, sq * c, hd, k, c, skv * c, hd, scores, skv, heads * sq * skv, sq * skv, sq, skv, hd, z, heads, alpha=scale)
    probs = ops.softmax_fwd(scores, pad8=True)  # [z, Sq, Skv] view, pitch kp
    del scores
    vt = tr(v, skv, kp)  # [z, hd, kp]
    o = torch.empty((b * sq, c), dtype=BF16, device=dev)
    _gemm(probs, kp, heads * sq * kp, sq * kp, vt, kp, heads * hd * kp, hd * kp, o, c, sq * c, hd, sq, hd, kp, z, heads)
    ctx.count(4 * b * sq * skv * c)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            do = tape.take(o)
            if do is None:
                return
            dp = torch.empty((z, sq, skv), dtype=F32, device=dev)
            _gemm(do, c, sq * c, hd, v, c, skv * c, hd, dp, skv, heads * sq * skv, sq * skv, sq, skv, hd, z, heads)
            ds = ops.softmax_bwd(probs, dp, scale)  # [z, Sq, Skv] view, pitch kp
            del dp
            kt = tr(k, skv, kp)
            dq = torch.empty((b * sq, c), dtype=BF16, device=dev)
            _gemm(ds, kp, heads * sq * kp, sq * kp, kt, kp, heads * hd * kp, hd * kp, dq, c, sq * c, hd, sq, hd, kp, z, heads)
            tape.put(q, dq)

            def tr_ss(src):  # [z][Sq][Skv] (pitch kp) -> [z][Skv][qp]
                out = torch.empty((z, skv, qp), dtype=BF16, device=dev)
                call("mi_transpose_bf16", ptr(src), kp, heads * sq * kp, sq * kp, ptr(out), qp, heads * skv * qp, skv * qp, sq, skv, z, heads)
                return out

            dst, qt = tr_ss(ds), tr(q, sq, qp)
            dk = torch.empty((b * skv, c), dtype=BF16, device=dev)
            _gemm(dst, qp, heads * skv * qp, skv * qp, qt, qp, heads * hd * qp, hd * qp, dk, c, skv * c, hd, skv, hd, qp, z, heads)
            pt, dot = tr_ss(probs), tr(do, sq, qp)
            dv = torch.empty((b * skv, c), dtype=BF16, device=dev)
            _gemm(pt, qp, heads * skv * qp, skv * qp, dot, qp, heads * hd * qp, hd * qp, dv, c, skv * c, hd, skv, hd, qp, z, heads)
            tape.put(k, dk), tape.put(v, dv)

        tape.record(bwd)
    return o


def spatial_transformer(ctx: Ctx, x, name, context, groups, eps, heads, layers):
    """SpatialTransformer.forward (UNet:314-342) with BasicTransformerBlock.forward (UNet:225-234) inlined:
    GroupNorm -> 1x1 proj_in -> [self-attention, cross-attention on `context`, GEGLU feed-forward] x layers -> 1x1 proj_out + x.
    context: bf16 [B * Sc, Cc] token matrix (or None: attn2 is a second self-attention, UNet:159); it is a constant of the step
    (a text / label encoder is trained through its own graph)."""
    b, d_, h_, w_, c = x.shape
    s = d_ * h_ * w_
    pre = name + "." if name else ""
    st = gn(ctx, x, pre + "norm", groups, eps)
    xn = gn_act(ctx, x, st, False)
    one, zero = (1, 1, 1), (0, 0, 0)
    t = _reshape(ctx, conv(ctx, xn, pre + "proj_in.conv", one, one, zero), (b * s, c))
    sc = context.shape[0] // b if context is not None else s
    for li in range(layers):
        blk = f"{pre}transformer_blocks.{li}"
        n1 = layernorm(ctx, t, blk + ".norm1")
        a1 = blk + ".attn1"
        o = mha(ctx, linear(ctx, n1, a1 + ".to_q", bias=False), linear(ctx, n1, a1 + ".to_k", bias=False),
                linear(ctx, n1, a1 + ".to_v", bias=False), b, s, s, heads)
        t = linear(ctx, o, a1 + ".to_out.0", res=t)
        n2 = layernorm(ctx, t, blk + ".norm2")
        a2 = blk + ".attn2"
        kv_src, kv_grad = (context, False) if context is not None else (n2, True)
        o = mha(ctx, linear(ctx, n2, a2 + ".to_q", bias=False), linear(ctx, kv_src, a2 + ".to_k", bias=False, need_dx=kv_grad),
                linear(ctx, kv_src, a2 + ".to_v", bias=False, need_dx=kv_grad), b, s, sc, heads)
        t = linear(ctx, o, a2 + ".to_out.0", res=t)
        n3 = layernorm(ctx, t, blk + ".norm3")
        t = linear(ctx, geglu(ctx, linear(ctx, n3, blk + ".ff.linear1")), blk + ".ff.linear2", res=t)
    return conv(ctx, _reshape(ctx, t, (b, d_, h_, w_, c)), pre + "proj_out.conv", one, one, zero, res=x)
