"""Raw (non-autograd) wrappers over the C ABI: one Python function per kernel family.

Tensors: activations are channels-last bf16 `[N, D, H, W, C]` (2-D nets use D = 1), fp32
for weights / statistics / gradients.  Outputs are allocated with torch.empty (caching
allocator, hipGraph-friendly); nothing here computes with torch ops.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import call, ptr

BF16, F32 = torch.bfloat16, torch.float32
import os as _os  # noqa: E402
# convs emit the per-channel sums of their output for the consuming GroupNorm (conv27 epilogue); 0: separate statistics pass
FUSE_GN_STATS = _os.environ.get("MI_FUSE_GN_STATS", "1") == "1"


def _vox(t):  # [N, D, H, W, C] -> (N, V, C)
    n, d, h, w, c = t.shape
    return n, d * h * w, c


def _cs(t):
    """channel stride (elements per voxel) of a channels-last activation (may be a channel slice of a wider buffer)."""
    assert t.dim() == 5
    if t.is_contiguous():  # also covers size-1 dims, whose strides torch does not normalise
        return t.shape[-1]
    assert t.stride(-1) == 1
    cs = t.stride(3)
    assert t.stride(2) == t.shape[3] * cs and t.stride(1) == t.shape[2] * t.stride(2) and t.stride(0) == t.shape[1] * t.stride(1), \
        "activation must be dense in N, D, H, W"
    return cs


# ----------------------------------------------------------------------------- layout
def to_channels_last(x_ncdhw: torch.Tensor) -> torch.Tensor:
    """fp32 NCDHW (or NCHW) -> bf16 NDHWC."""
    x = x_ncdhw.contiguous()
    if x.dim() == 4:
        x = x.unsqueeze(2)
    n, c, d, h, w = x.shape
    out = torch.empty((n, d, h, w, c), dtype=BF16, device=x.device)
    call("mi_ncdhw_f32_to_ndhwc_bf16", ptr(x), ptr(out), n, c, d * h * w)
    return out


def to_channels_first(x_cl: torch.Tensor, spatial_dims: int = 3) -> torch.Tensor:
    """bf16 NDHWC -> fp32 NCDHW (NCHW when spatial_dims == 2)."""
    assert x_cl.is_contiguous()
    n, d, h, w, c = x_cl.shape
    out = torch.empty((n, c, d, h, w), dtype=F32, device=x_cl.device)
    call("mi_ndhwc_bf16_to_ncdhw_f32", ptr(x_cl), ptr(out), n, c, d * h * w)
    return out.squeeze(2) if spatial_dims == 2 else out


def add(a, b):
    a, b = dense(a), dense(b)
    assert a.shape == b.shape
    out = torch.empty_like(a)
    call("mi_add_bf16", ptr(a), ptr(b), ptr(out), a.numel())
    return out


def concat_channels(a, b):
    n, v, ca = _vox(a)
    cb = b.shape[-1]
    out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=BF16, device=a.device)
    call("mi_copy_channels", ptr(a), _cs(a), 0, ptr(out), ca + cb, 0, ca, n * v)
    call("mi_copy_channels", ptr(b), _cs(b), 0, ptr(out), ca + cb, ca, cb, n * v)
    return out


def slice_channels(x, c0, nc):
    n, v, c = _vox(x)
    out = torch.empty(x.shape[:-1] + (nc,), dtype=BF16, device=x.device)
    call("mi_copy_channels", ptr(x), _cs(x), c0, ptr(out), nc, 0, nc, n * v)
    return out


def dense(t):
    """A channel-slice view of a wider channels-last buffer (the gradient hand-off of an in-place concat) as a dense tensor; dense
    tensors pass through.  Our copy kernel, not aten."""
    return t if t.is_contiguous() else slice_channels(t, 0, t.shape[-1])


def upsample_nearest(x, f):
    x = dense(x)
    n, d, h, w, c = x.shape
    out = torch.empty((n, d * f[0], h * f[1], w * f[2], c), dtype=BF16, device=x.device)
    call("mi_upsample_nearest_fwd", ptr(x), ptr(out), n, d, h, w, c, f[0], f[1], f[2])
    return out


def upsample_nearest_bwd(dy, f):
    n, d, h, w, c = dy.shape
    out = torch.empty((n, d // f[0], h // f[1], w // f[2], c), dtype=BF16, device=dy.device)
    call("mi_upsample_nearest_bwd", ptr(dy), ptr(out), n, d // f[0], h // f[1], w // f[2], c, f[0], f[1], f[2])
    return out


def _pool_out(dims, kernel, stride):
    return tuple((d - k) // s + 1 for d, k, s in zip(dims, kernel, stride))


def avg_pool(x, kernel, stride):
    """nn.AvgPool{2,3}d(kernel_size, stride) (no padding, floor mode) on NDHWC bf16."""
    x = dense(x)  # (a skip living in its concatenation buffer is a channel-slice view)
    n, d, h, w, c = x.shape
    od, oh, ow = _pool_out((d, h, w), kernel, stride)
    out = torch.empty((n, od, oh, ow, c), dtype=BF16, device=x.device)
    arr = lambda v: (C.c_int * 3)(*v)
    call("mi_avgpool_fwd", ptr(x), ptr(out), n, d, h, w, c, arr(kernel), arr(stride))
    return out


def avg_pool_bwd(dy, in_dims, kernel, stride):
    n, c = dy.shape[0], dy.shape[-1]
    assert dy.is_contiguous() and tuple(dy.shape[1:4]) == _pool_out(in_dims, kernel, stride)
    dx = torch.empty((n,) + tuple(in_dims) + (c,), dtype=BF16, device=dy.device)
    arr = lambda v: (C.c_int * 3)(*v)
    call("mi_avgpool_bwd", ptr(dy), ptr(dx), n, in_dims[0], in_dims[1], in_dims[2], c, arr(kernel), arr(stride))
    return dx


# ----------------------------------------------------------------------------- GroupNorm
_ws_cache: dict = {}


def _workspace(nbytes, device):
    key = (device.index, "gn")
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class GNStats:
    """Per-(n, channel) scale / shift and per-(n, group) mean / rstd of one GroupNorm.  `pending`: the statistics of a SMALL tensor are
    not computed yet -- gn_apply() will produce them together with the activated tensor in one launch (mi_gn_small_fwd); any other
    reader of the records gets them from the plain statistics pass first."""
    __slots__ = ("_ss", "_mr", "groups", "name", "_pending")

    def __init__(self, scale_shift, mean_rstd, groups, pending=None):
        self._ss, self._mr, self.groups, self._pending = scale_shift, mean_rstd, groups, pending

    def _force(self):
        if self._pending is not None:
            x, eps, gamma, beta = self._pending
            self._pending = None
            st = gn_stats(x, self.groups, eps, gamma, beta, allow_small=False)
            self._ss, self._mr = st._ss, st._mr

    @property
    def scale_shift(self):
        self._force()
        return self._ss

    @property
    def mean_rstd(self):
        self._force()
        return self._mr


# A/B knob, default OFF: single-launch GroupNorm (statistics + coefficients + apply) for tensors a workgroup per group can hold.  Measured
# (round 3, profiles/r03gs_ab_gn_small_fwd.log): 21.34 ms/step either way -- the launch takes 16.8 us (32 workgroups pulling 16-byte
# pieces at a 512-byte stride through 32 of the 256 CUs) where the three chip-wide launches it replaces take ~18 us with their boundaries.
GN_SMALL = os.environ.get("MI_GN_SMALL", "0") == "1"


def gn_stats(x, groups, eps, gamma, beta, allow_small=True) -> GNStats:
    n, v, c = _vox(x)
    if allow_small and GN_SMALL and c % groups == 0 and _lib.call_raw("mi_gn_small_supported", n, v, c, groups):
        return GNStats(None, None, groups, pending=(x, float(eps), gamma, beta))
    ss = torch.empty((n, c, 2), dtype=F32, device=x.device)
    mr = torch.empty((n, groups, 2), dtype=F32, device=x.device)
    nb = _lib.call_raw("mi_gn_workspace_bytes", n, v, c)
    ws = _workspace(nb, x.device)
    call("mi_gn_stats", ptr(x), _cs(x), n, v, c, groups, float(eps), ptr(gamma), ptr(beta), ptr(ss), ptr(mr), ptr(ws), ws.numel())
    return GNStats(ss, mr, groups)


class ChannelSums:
    """Per-channel (sum, sum of squares) partials of a tensor, [N][C][chunks][2] fp32, emitted by the conv that produced it."""
    __slots__ = ("partial", "chunks", "channels")

    def __init__(self, partial, chunks, channels):
        self.partial, self.chunks, self.channels = partial, chunks, channels


def gn_stats_from_sums(a: ChannelSums, b: ChannelSums | None, n, v, groups, eps, gamma, beta) -> GNStats | None:
    """GroupNorm statistics of a tensor whose channels are those of `a` followed by those of `b` (or of `a` alone) without reading
    the tensor."""
    c = a.channels + (b.channels if b is not None else 0)
    dev = a.partial.device
    ss = torch.empty((n, c, 2), dtype=F32, device=dev)
    mr = torch.empty((n, groups, 2), dtype=F32, device=dev)
    call("mi_gn_stats_from_partial", ptr(a.partial), a.chunks, a.channels, ptr(b.partial) if b is not None else None,
         b.chunks if b is not None else 0, b.channels if b is not None else 0, n, v, groups, float(eps), ptr(gamma), ptr(beta), ptr(ss), ptr(mr))
    return GNStats(ss, mr, groups)


def gn_apply(x, st: GNStats, silu: bool):
    n, v, c = _vox(x)
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    if st._pending is not None and st._pending[0] is x:
        _, eps, gamma, beta = st._pending
        st._pending = None
        st._ss = torch.empty((n, c, 2), dtype=F32, device=x.device)
        st._mr = torch.empty((n, st.groups, 2), dtype=F32, device=x.device)
        call("mi_gn_small_fwd", ptr(x), _cs(x), ptr(y), c, n, v, c, st.groups, eps, ptr(gamma), ptr(beta), ptr(st._ss), ptr(st._mr), int(silu))
        return y
    call("mi_gn_apply", ptr(x), _cs(x), ptr(st.scale_shift), ptr(y), c, n, v, c, int(silu))
    return y


GN_FUSED_REPLICAS = 16  # MI_GN_FUSED_REPLICAS of include/medimgen_hip.h


def gn_bwd(g, x, st: GNStats, gamma, silu: bool, dgamma, dbeta, add=None, add2=None, sums=None):
    """g = dL/d(act(GN(x))) -> dL/dx (+ add + add2: other pending branches of x's gradient, views allowed); dgamma/dbeta (fp32) are
    accumulated in place.  sums: a ZEROED fp64 buffer of GN_FUSED_REPLICAS * n * c * 2 elements (engine.Ctx.zeros64) selects the two-launch form
    (mi_gn_bwd_fused: atomics instead of the finalize launch)."""
    if add is None and add2 is not None:
        add, add2 = add2, None
    n, v, c = _vox(x)
    dx = torch.empty(x.shape, dtype=BF16, device=x.device)
    if sums is not None:
        assert sums.dtype == torch.float64 and sums.numel() >= GN_FUSED_REPLICAS * n * c * 2
        call("mi_gn_bwd_fused", ptr(g), _cs(g), ptr(x), _cs(x), n, v, c, st.groups, ptr(gamma), ptr(st.scale_shift), ptr(st.mean_rstd),
             int(silu), ptr(add), _cs(add) if add is not None else 0, ptr(add2), _cs(add2) if add2 is not None else 0, ptr(dx), c,
             ptr(dgamma), ptr(dbeta), ptr(sums))
        return dx
    coef = torch.empty((n, c, 3), dtype=F32, device=x.device)
    nb = _lib.call_raw("mi_gn_workspace_bytes", n, v, c)
    ws = _workspace(nb, x.device)
    call("mi_gn_bwd", ptr(g), _cs(g), ptr(x), _cs(x), n, v, c, st.groups, ptr(gamma), ptr(st.scale_shift), ptr(st.mean_rstd),
         int(silu), ptr(add), _cs(add) if add is not None else 0, ptr(add2), _cs(add2) if add2 is not None else 0, ptr(dx), c, ptr(dgamma),
         ptr(dbeta), ptr(coef), ptr(ws), ws.numel())
    return dx


# ----------------------------------------------------------------------------- convolution
class ConvPlan:
    """Opaque mi_conv_plan handle for one (shape, kernel, stride, padding) instance."""

    def __init__(self, n, dims, cin, cout, kernel, stride, padding):
        self.n, self.dims, self.cin, self.cout = n, tuple(dims), cin, cout
        self.kernel, self.stride, self.padding = tuple(kernel), tuple(stride), tuple(padding)
        h = C.c_void_p()
        arr = lambda v: (C.c_int * 3)(*v)
        _lib.call_raw("mi_conv_plan_create", C.byref(h), n, dims[0], dims[1], dims[2], cin, cout, arr(kernel), arr(stride), arr(padding))
        self.handle = h
        od = (C.c_int * 3)()
        _lib.call_raw("mi_conv_plan_out_dims", self.handle, od)
        self.out_dims = tuple(od)
        self.stats_chunks = int(_lib.call_raw("mi_conv_fwd_stats_chunks", self.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.call_raw("mi_conv_plan_destroy", self.handle)
                self.handle = None
        except Exception:
            pass

    def pack(self, weight_f32):
        assert weight_f32.dtype == F32 and weight_f32.is_contiguous()
        call("mi_conv_pack_weights", self.handle, ptr(weight_f32))

    def fwd(self, x, st: GNStats | None = None, silu=False, addvec=None, res=None, out=None, want_sums=False):
        """out: optional destination, a channel-slice view [N, D, H, W, Cout] of a wider channels-last buffer (the conv writes
        straight into the skip-concat buffer of its consumer instead of being copied there).  want_sums: also return the
        ChannelSums of y when this plan's kernel can emit them (else None): (y, sums)."""
        n, d, h, w, c = x.shape
        assert (n, (d, h, w), c) == (self.n, self.dims, self.cin), f"plan/input mismatch {x.shape} vs {self.n, self.dims, self.cin}"
        if out is None:
            y = torch.empty((n,) + self.out_dims + (self.cout,), dtype=BF16, device=x.device)
        else:
            assert out.shape == (n,) + self.out_dims + (self.cout,) and out.dtype == BF16
            y = out
        av_stride = 0
        if addvec is not None and addvec.dim() == 2:  # [N, Cout] rows, possibly a column slice of a wider matrix
            assert addvec.shape == (n, self.cout) and addvec.stride(1) == 1
            av_stride = addvec.stride(0)
        sums = None
        if want_sums and FUSE_GN_STATS and st is None and self.stats_chunks > 0 and (_cs(x) % 8 == 0 or (self.cin == 1 and res is None)):
            sums = ChannelSums(torch.empty((n, self.cout, self.stats_chunks, 2), dtype=F32, device=x.device), self.stats_chunks, self.cout)
        call("mi_conv_fwd", self.handle, ptr(x), _cs(x), ptr(st.scale_shift) if st is not None else None, int(silu), ptr(addvec), av_stride,
             ptr(res), _cs(res) if res is not None else 0, ptr(y), _cs(y), ptr(sums.partial) if sums is not None else None)
        return (y, sums) if want_sums else y

    def dgrad(self, dy):
        assert dy.shape == (self.n,) + self.out_dims + (self.cout,)
        dx = torch.empty((self.n,) + self.dims + (self.cin,), dtype=BF16, device=dy.device)
        call("mi_conv_dgrad", self.handle, ptr(dy), _cs(dy), ptr(dx), self.cin)
        return dx

    def wgrad(self, x, dy, dweight_f32, st: GNStats | None = None, silu=False, colsum=None):
        """dweight += wgrad; colsum (optional fp32) += column sums of dy: an [N, Cout] view (row pitch honoured) receives them
        per image, a [Cout] vector their sum over the batch (the bias gradient)."""
        assert dweight_f32.dtype == F32 and dweight_f32.is_contiguous()
        cs_stride = 0
        if colsum is not None:
            assert colsum.dtype == F32 and colsum.stride(-1) == 1
            if colsum.dim() == 2:
                assert colsum.shape == (self.n, self.cout)
                cs_stride = colsum.stride(0)
            else:
                assert colsum.shape == (self.cout,)
        call("mi_conv_wgrad", self.handle, ptr(x), _cs(x), ptr(st.scale_shift) if st is not None else None, int(silu), ptr(dy), _cs(dy),
             ptr(dweight_f32), ptr(colsum), cs_stride)


class UpConvPlan(ConvPlan):
    """Upsample.forward as one op (mi_upconv_plan_create): nearest x2 on all three axes, then the k3 s1 p1 conv; x is the COARSE
    tensor [N, D, H, W, Cin], y [N, 2D, 2H, 2W, Cout].  Same methods as ConvPlan (dgrad returns the coarse dx)."""

    def __init__(self, n, dims, cin, cout):
        self.n, self.dims, self.cin, self.cout = n, tuple(dims), cin, cout
        self.kernel, self.stride, self.padding = (3, 3, 3), (1, 1, 1), (1, 1, 1)
        h = C.c_void_p()
        _lib.call_raw("mi_upconv_plan_create", C.byref(h), n, dims[0], dims[1], dims[2], cin, cout)
        self.handle = h
        self.out_dims = tuple(2 * d for d in dims)
        self.stats_chunks = 0


def colsum(x, out=None, accumulate=False, merge_batch=False):
    """out[n, c] (+)= sum over voxels; `out` may be a column slice of a wider fp32 matrix.  merge_batch: sum over n too."""
    n, v, c = _vox(x)
    assert x.is_contiguous()
    if merge_batch:
        n, v = 1, n * v
    if out is None:
        out = torch.empty((n, c), dtype=F32, device=x.device)
    stride = out.stride(0) if out.dim() == 2 else c
    call("mi_colsum_bf16", ptr(x), ptr(out), stride, n, v, c, int(accumulate))
    return out


def add_f32_(y, x):
    """y += x for small fp32 matrices / vectors (row pitches honoured)."""
    y2 = y if y.dim() == 2 else y.unsqueeze(0)
    if x.dim() == 1:  # broadcast one row over every row of y
        assert x.shape[0] == y2.shape[1] and x.stride(0) == 1
        ldx, xp = 0, x
    else:
        assert x.shape == y2.shape and x.stride(1) == 1
        ldx, xp = x.stride(0), x
    assert y2.stride(1) == 1
    call("mi_add_f32_2d", ptr(xp), ldx, ptr(y2), y2.stride(0), y2.shape[0], y2.shape[1])
    return y


def zero_f32_2d_(x):
    """zero a small fp32 matrix that may be a column slice of a wider one (hipMemset2DAsync, no torch kernel)."""
    assert x.dim() == 2 and x.stride(1) == 1
    call("mi_zero_f32_2d", ptr(x), x.stride(0), x.shape[0], x.shape[1])
    return x


def sum_rows_f32(x, out, accumulate=True):
    assert x.dim() == 2 and x.stride(1) == 1 and out.is_contiguous()
    call("mi_sum_rows_f32", ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(out), int(accumulate))
    return out


# ----------------------------------------------------------------------------- GEMM family
def gemm_nt(a, b, *, bias=None, res=None, alpha=1.0, out=None, out_f32=False, accumulate=False):
    """out[z] = alpha * a[z] @ b[z]^T (+bias) (+res).  a: [Z?, M, K], b: [Z?, N, K] (last dim contiguous, arbitrary row pitch)."""
    a3, b3 = (a if a.dim() == 3 else a.unsqueeze(0)), (b if b.dim() == 3 else b.unsqueeze(0))
    z, m, k = a3.shape
    n = b3.shape[1]
    assert b3.shape[2] == k and a3.stride(2) == 1 and b3.stride(2) == 1
    zb = b3.shape[0]
    assert zb in (1, z)
    if out is None:
        out = torch.empty((z, m, n), dtype=F32 if out_f32 else BF16, device=a.device)
        out_v = out
    else:
        out_v = out if out.dim() == 3 else out.unsqueeze(0)
    r3 = None
    if res is not None:
        r3 = res if res.dim() == 3 else res.unsqueeze(0)
        assert r3.stride(2) == 1
    call("mi_gemm_nt_bf16", ptr(a3), a3.stride(1), a3.stride(0), 0, ptr(b3), b3.stride(1), b3.stride(0) if zb == z else 0, 0,
         ptr(out_v), out_v.stride(1), out_v.stride(0), 0, ptr(bias), ptr(r3), r3.stride(1) if r3 is not None else 0,
         r3.stride(0) if r3 is not None else 0, 0, m, n, k, z, 1, float(alpha), int(out_v.dtype == F32), int(accumulate))
    return out if a.dim() == 3 or out.dim() == 2 else out[0]


def transpose(x):
    """[Z?, R, C] bf16 (last dim contiguous) -> [Z?, C, R] contiguous."""
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    z, r, c = x3.shape
    out = torch.empty((z, c, r), dtype=BF16, device=x.device)
    call("mi_transpose_bf16", ptr(x3), x3.stride(1), x3.stride(0), 0, ptr(out), r, c * r, 0, r, c, z, 1)
    return out if x.dim() == 3 else out[0]


def softmax_fwd(scores_f32, pad8=False):
    """bf16 row softmax of dense fp32 scores.  pad8: the result's row pitch is the column count rounded up to a multiple of 8, pad
    columns zero (returned as a [..., cols] view of the padded buffer) -- what the NT GEMM needs when it reduces over this axis."""
    s = scores_f32.contiguous()
    cols = s.shape[-1]
    ld = (cols + 7) // 8 * 8 if pad8 else cols
    out = torch.empty(s.shape[:-1] + (ld,), dtype=BF16, device=s.device)
    call("mi_softmax_fwd", ptr(s), ptr(out), s.numel() // cols, cols, ld)
    return out[..., :cols]


def softmax_bwd(probs, dprobs_f32, scale):
    """probs: as returned by softmax_fwd (possibly a view of a padded buffer: the result has the same pitch, pad columns zero)."""
    cols, ld = probs.shape[-1], probs.stride(-2)
    assert probs.stride(-1) == 1 and dprobs_f32.is_contiguous()
    out = torch.empty(probs.shape[:-1] + (ld,), dtype=BF16, device=probs.device)
    call("mi_softmax_bwd", ptr(probs), ptr(dprobs_f32), ptr(out), probs.numel() // cols, cols, ld, float(scale))
    return out[..., :cols]


# ----------------------------------------------------------------------------- small ops
def timestep_embedding(t_i64, dim, max_period=10000.0):
    out = torch.empty((t_i64.shape[0], dim), dtype=F32, device=t_i64.device)
    call("mi_timestep_embedding", ptr(t_i64), ptr(out), t_i64.shape[0], dim, float(max_period))
    return out


def silu_f32(x):
    y = torch.empty_like(x)
    call("mi_silu_f32", ptr(x), ptr(y), x.numel())
    return y


def silu_bwd_f32(x, dy):
    dx = torch.empty_like(x)
    call("mi_silu_bwd_f32", ptr(x), ptr(dy), ptr(dx), x.numel())
    return dx


def cast_bf16(x_f32):
    assert x_f32.is_contiguous() and x_f32.dtype == F32
    out = torch.empty(x_f32.shape, dtype=BF16, device=x_f32.device)
    call("mi_cast_f32_to_bf16", ptr(x_f32), ptr(out), x_f32.numel())
    return out


def cast_f32(x_bf16, out=None, accumulate=False):
    if out is None:
        out = torch.empty(x_bf16.shape, dtype=F32, device=x_bf16.device)
    call("mi_cast_bf16_to_f32", ptr(x_bf16), ptr(out), x_bf16.numel(), int(accumulate))
    return out
