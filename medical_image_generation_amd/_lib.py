"""ctypes binding of libmedimgen_hip.so (the C ABI declared in include/medimgen_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
PyTorch is used only as the owner of device memory and streams; every pointer handed to the
library is `tensor.data_ptr()`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_LIB_PATH") or os.path.join(_HERE, "libmedimgen_hip.so")  # MI_LIB_PATH: ablation builds (csrc/Makefile `diag`)

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (return type is int unless listed in _RET)
_SIGS = {
    "mi_abi_version": [],
    "mi_ncdhw_f32_to_ndhwc_bf16": [_p, _p, _i, _i, _l, _p],
    "mi_ndhwc_bf16_to_ncdhw_f32": [_p, _p, _i, _i, _l, _p],
    "mi_cast_f32_to_bf16": [_p, _p, _l, _p],
    "mi_cast_bf16_to_f32": [_p, _p, _l, _i, _p],
    "mi_add_bf16": [_p, _p, _p, _l, _p],
    "mi_copy_channels": [_p, _i, _i, _p, _i, _i, _i, _l, _p],
    "mi_upsample_nearest_fwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_upsample_nearest_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_space_to_depth": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_depth_to_space": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_gn_workspace_bytes": [_i, _l, _i],
    "mi_gn_stats": [_p, _i, _i, _l, _i, _i, _f, _p, _p, _p, _p, _p, _l, _p],
    "mi_gn_stats_from_partial": [_p, _i, _i, _p, _i, _i, _i, _l, _i, _f, _p, _p, _p, _p, _p],
    "mi_gn_apply": [_p, _i, _p, _p, _i, _i, _l, _i, _i, _p],
    "mi_gn_bwd": [_p, _i, _p, _i, _i, _l, _i, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _l, _p],
    "mi_gn_bwd_fused": [_p, _i, _p, _i, _i, _l, _i, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p],
    "mi_conv_plan_create": [C.POINTER(_p), _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "mi_upconv_plan_create": [C.POINTER(_p), _i, _i, _i, _i, _i, _i],
    "mi_conv_plan_destroy": [_p],
    "mi_conv_plan_out_dims": [_p, C.POINTER(_i)],
    "mi_conv_pack_weights": [_p, _p, _p],
    "mi_conv_pack_batch_create": [C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), _i],
    "mi_conv_pack_batch_run": [_p, _p],
    "mi_conv_pack_batch_destroy": [_p],
    "mi_conv_fwd": [_p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _p],
    "mi_conv_fwd_stats_chunks": [_p],
    "mi_conv_dgrad": [_p, _p, _i, _p, _i, _p],
    "mi_conv_wgrad": [_p, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p],
    "mi_linear_wgrad_bf16": [_p, _i, _i, _p, _i, _i, _l, _p, _p, _p],
    "mi_colsum_bf16": [_p, _p, _i, _i, _l, _i, _i, _p],
    "mi_add_f32_2d": [_p, _i, _p, _i, _i, _i, _p],
    "mi_sum_rows_f32": [_p, _i, _i, _i, _p, _i, _p],
    "mi_zero_f32_2d": [_p, _i, _i, _i, _p],
    "mi_gemm_nt_bf16": [_p, _i, _l, _l, _p, _i, _l, _l, _p, _i, _l, _l, _p, _p, _i, _l, _l, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "mi_transpose_bf16": [_p, _i, _l, _l, _p, _i, _l, _l, _i, _i, _i, _i, _p],
    "mi_softmax_fwd": [_p, _p, _l, _i, _i, _p],
    "mi_softmax_bwd": [_p, _p, _p, _l, _i, _i, _f, _p],
    "mi_attn_supported": [_i, _i],
    "mi_attn_workspace_bytes": [_i, _i, _i, _i],
    "mi_attn_fwd": [_p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _l, _p],
    "mi_attn_bwd": [_p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _l, _p],
    "mi_timestep_embedding": [_p, _p, _i, _i, _f, _p],
    "mi_silu_f32": [_p, _p, _l, _p],
    "mi_silu_bwd_f32": [_p, _p, _p, _l, _p],
    "mi_logvar_to_sigma_fwd": [_p, _p, _l, _p],
    "mi_logvar_to_sigma_bwd": [_p, _p, _p, _p, _l, _p],
    "mi_embedding_add": [_p, _p, _p, _i, _i, _p],
    "mi_embedding_bwd": [_p, _p, _p, _i, _i, _p],
    "mi_layernorm_fwd": [_p, _i, _p, _p, _p, _i, _p, _l, _i, _f, _p],
    "mi_layernorm_bwd": [_p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _l, _i, _p],
    "mi_geglu_fwd": [_p, _p, _l, _i, _p],
    "mi_geglu_bwd": [_p, _p, _p, _l, _i, _p],
    "mi_avgpool_fwd": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "mi_avgpool_bwd": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "mi_crop_pad": [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _f, _i, _f, _i, _p],
    "mi_gn_small_supported": [_i, _l, _i, _i],
    "mi_gn_small_fwd": [_p, _i, _p, _i, _i, _l, _i, _i, _f, _p, _p, _p, _p, _i, _p],
    "mi_aug_stats_workspace_bytes": [],
    "mi_aug_plane_stats": [_p, _l, _p, _p, _p],
    "mi_aug_pointwise": [_p, _l, _i, _f, _p, _p, _p, _p],
    "mi_aug_blur_axis": [_p, _p, _i, _i, _i, _i, _p, _i, _p],
    "mi_aug_lowres": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mi_aug_affine_sample": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "mi_im2col3d": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_col2im3d": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "mi_disc_pack_weights": [_p, _p, _p, _i, _i, _i, _i, _p],
    "mi_disc_wgrad_unpack": [_p, _p, _i, _i, _i, _p],
    "mi_leaky_relu_fwd": [_p, _p, _l, _f, _p],
    "mi_leaky_relu_bwd": [_p, _p, _p, _l, _f, _p],
    "mi_ls_gan_loss": [_p, _i, _l, _f, _f, _p, _p, _f, _p],
    "mi_bn_running_update": [_p, _p, _p, _p, _i, _f, _f, _l, _p],
    "mi_qsample": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _l, _i, _p],
    "mi_ddpm_step": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _i, _p],
    "mi_mse_fwd_bwd": [_p, _p, _p, _p, _i, _i, _l, _f, _p],
    "mi_l1_fwd_bwd": [_p, _p, _p, _p, _i, _i, _l, _i, _p],
    "mi_reparam_kl_fwd": [_p, _p, _p, _p, _p, _i, _i, _l, _f, _p],
    "mi_reparam_kl_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _l, _f, _p],
    "mi_sumsq_f32": [_p, _l, _p, _i, _p],
    "mi_clip_grad_by_norm": [_p, _l, _p, _f, _p],
    "mi_adam_step": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p, _f, _f, _p, _p],
    "mi_axpy_f32": [_p, _p, _f, _l, _p],
    "mi_scale_f32": [_p, _f, _l, _p],
}
_RET = {"mi_gn_workspace_bytes": _l, "mi_attn_workspace_bytes": _l, "mi_aug_stats_workspace_bytes": _l}
_NOCHECK = {"mi_abi_version", "mi_gn_small_supported", "mi_gn_workspace_bytes", "mi_aug_stats_workspace_bytes", "mi_attn_supported", "mi_attn_workspace_bytes", "mi_conv_fwd_stats_chunks"}

_lib = None
# Version of the C ABI this binding was written against (csrc/api.hip: mi_abi_version).  Entry points have changed their argument
# lists under unchanged names between versions, and *.so files are not tracked by git: a stale library (or an MI_LIB_PATH pointing at
# an old ablation build) resolves every symbol and then reads shifted arguments.  load() refuses it.
ABI_VERSION = 8


def exported_symbols() -> list[str]:
    return sorted(_SIGS)


def load():
    """dlopen the library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU / PyTorch fallback for the HIP path)")
        lib = C.CDLL(LIB_PATH)
        try:
            lib.mi_abi_version.restype = _i
            have = int(lib.mi_abi_version())
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} is a stale library (C ABI {have}, this binding needs {ABI_VERSION}): rebuild it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
        for name, args in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.argtypes = args
            fn.restype = _RET.get(name, _i)
        _lib = lib
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class HipError(RuntimeError):
    pass


_ERRS = {10001: "MI_ERR_BAD_ARG (host-side shape/alignment check failed)", 10002: "MI_ERR_UNSUPPORTED"}


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "medimgen HIP ops take device tensors only"
    return t.data_ptr()


_profile = None


class profile_calls:
    """Context manager: HIP events on the launch stream around every library call made inside it (eager mode only -- events cannot
    be recorded into a capturing stream this way); summary() -> {entry point: GPU milliseconds}.  Measurement aid for bench.py."""

    def __enter__(self):
        global _profile
        self.records = []
        _profile = self
        return self

    def __exit__(self, *exc):
        global _profile
        _profile = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


def call(name: str, *args):
    """Invoke `name` on the current stream (appended as the last argument) and check the status."""
    fn = getattr(load(), name)
    prof = _profile
    if prof is not None:
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
    rc = fn(*args, stream_ptr())
    if prof is not None:
        e1.record(st)
        prof.records.append((name, e0, e1))
    if rc != 0:
        raise HipError(f"{name} failed: {_ERRS.get(rc, f'hipError_t {rc}')}")


def call_raw(name: str, *args):
    fn = getattr(load(), name)
    rc = fn(*args)
    if name not in _NOCHECK and rc != 0:
        raise HipError(f"{name} failed: {_ERRS.get(rc, f'hipError_t {rc}')}")
    return rc
