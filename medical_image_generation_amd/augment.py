"""Training-time augmentation on the GPU: `define_nnunet_transformations(params, validation)` of
medimgen/data_processing.py:748-859 for patches that live in HBM.

The reference composes batchgeneratorsv2 transforms and runs them on the CPU in DataLoader workers; here the same list is a chain of
HIP launches (`csrc/augment.hip`) on the patch `GpuPatchLoader` has just cut, so an augmented batch never leaves the device.  The
function takes the reference's `params` dictionary (keys of MedicalDataset's augmentation setting, DATA:399-424: rotation, rot_for_da,
scaling, scaling_range, gaussian_noise, gaussian_blur, brightness, brightness_range, contrast, contrast_range, low_resolution, gamma,
gamma_range, mirror_axes, patch_size, dummy_2d) and returns a `ComposeTransforms` that is called like the reference's:
`chain(image=tensor[C, *spatial])["image"]` (DATA:530-533), or on a whole batch `[B, C, *spatial]` as `GpuPatchLoader(transform=chain)`.

PARITY UNPINNED: batchgeneratorsv2 is not under /root/reference and not installed.  What each transform computes, its probabilities and
its parameter ranges are restated from that package's published transforms (the arithmetic: oracle/data.py, against which the kernels
are tested with torch's own interpolate / grid_sample / conv / pad); the ORDER of the random draws is this module's own (all scalar
draws on numpy's global generator, the noise field from torch's device generator), so a seeded run is reproducible here but does not
replay the reference's stream.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from ._lib import call, call_raw, ptr

F32 = torch.float32
SCALE, CONTRAST, GAMMA, RESTORE_STATS, ADD_NOISE, CLAMP01 = range(6)  # MI_AUG_* of include/medimgen_hip.h


# ------------------------------------------------------------------ scalar samplers (host; a few numbers per sample)
def sample_scalar(spec, **kw):
    """A number, a (lo, hi) pair (uniform) or a callable (called with **kw), like the package's sample_scalar."""
    if callable(spec):
        return float(spec(**kw))
    if isinstance(spec, (tuple, list)):
        return float(np.random.uniform(spec[0], spec[1]))
    return float(spec)


class BGContrast:
    """The (lo, hi) -> multiplier sampler nnU-Net uses for brightness / contrast / gamma: with probability 0.5 (and lo < 1) a value
    below 1 from U(lo, 1), otherwise from U(max(lo, 1), hi)."""

    def __init__(self, contrast_range):
        self.contrast_range = tuple(contrast_range)

    def __call__(self, **_):
        lo, hi = self.contrast_range
        if np.random.random() < 0.5 and lo < 1:
            return np.random.uniform(lo, 1)
        return np.random.uniform(max(lo, 1), hi)


def gaussian_taps(sigma: float, truncate: float = 6.0) -> np.ndarray:
    """Normalised 1-D Gaussian of odd length round(sigma * truncate + 0.5) (made odd by adding one)."""
    k = int(round(sigma * truncate + 0.5))
    k += 1 - k % 2
    ax = np.arange(k, dtype=np.float32) - k // 2
    w = np.exp(-0.5 * (ax / np.float32(sigma)) ** 2).astype(np.float32)
    return w / w.sum(dtype=np.float32)


def affine_matrix(angles, scales) -> np.ndarray:
    """3x3 source-from-output matrix of the spatial transform: (Rz Ry Rx S)^T acting on centred (d, h, w) coordinates; `angles[i]`
    rotates about axis i, `scales[i]` > 1 samples a larger extent (the content shrinks)."""
    a0, a1, a2 = (float(a) for a in angles)
    rx = np.array([[1, 0, 0], [0, math.cos(a0), -math.sin(a0)], [0, math.sin(a0), math.cos(a0)]])
    ry = np.array([[math.cos(a1), 0, math.sin(a1)], [0, 1, 0], [-math.sin(a1), 0, math.cos(a1)]])
    rz = np.array([[math.cos(a2), -math.sin(a2), 0], [math.sin(a2), math.cos(a2), 0], [0, 0, 1]])
    return (rz @ ry @ rx @ np.diag([float(s) for s in scales])).T.astype(np.float32)


# ------------------------------------------------------------------ per-plane primitives (one launch each)
class _Scratch:
    """Device buffers a chain reuses: two statistics records, the reduction workspace, scratch planes."""

    def __init__(self):
        self.stats = self.work = None
        self.planes = {}

    def bind(self, device):
        if self.stats is None or self.stats.device != device:
            self.stats = torch.empty(2, 4, dtype=F32, device=device)
            self.work = torch.empty(int(call_raw("mi_aug_stats_workspace_bytes")) // 4, dtype=F32, device=device)
            self.planes = {}

    def plane(self, slot, like):
        key = (slot, tuple(like.shape))
        if key not in self.planes:
            self.planes[key] = torch.empty_like(like)
        return self.planes[key]


def _dhw(plane):
    s = tuple(plane.shape)
    return (1,) * (3 - len(s)) + s


def _check(plane):
    if not plane.is_cuda:
        raise RuntimeError("medical_image_generation_amd.augment runs on the GPU: the patch must be a 'cuda' tensor")
    if plane.dtype != F32 or not plane.is_contiguous() or not 2 <= plane.dim() <= 3:
        raise ValueError("a plane is a contiguous fp32 [D, H, W] or [H, W] tensor")


def plane_stats(plane, out, work):
    call("mi_aug_plane_stats", ptr(plane), plane.numel(), ptr(out), ptr(work))


def pointwise(plane, op, p0=0.0, stats_a=None, stats_b=None, aux=None):
    call("mi_aug_pointwise", ptr(plane), plane.numel(), op, float(p0), ptr(stats_a), ptr(stats_b), ptr(aux))


def blur_axis(src, dst, axis, taps):
    d, h, w = _dhw(src)
    t = np.ascontiguousarray(taps, dtype=np.float32)
    call("mi_aug_blur_axis", ptr(src), ptr(dst), d, h, w, axis + 3 - src.dim(), t.ctypes.data_as(C.c_void_p), len(t))


def lowres(src, dst, low_shape):
    d, h, w = _dhw(src)
    l = (1,) * (3 - len(low_shape)) + tuple(int(v) for v in low_shape)
    call("mi_aug_lowres", ptr(src), ptr(dst), d, h, w, l[0], l[1], l[2])


def affine_sample(src, dst, matrix):
    d, h, w = _dhw(src)
    od, oh, ow = _dhw(dst)
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    call("mi_aug_affine_sample", ptr(src), ptr(dst), d, h, w, od, oh, ow, m.ctypes.data_as(C.c_void_p))


# ------------------------------------------------------------------ the transforms (names of the classes the reference composes)
class BasicTransform:
    def bind(self, scratch):
        self.scratch = scratch
        return self

    def __call__(self, image):
        raise NotImplementedError


class RandomTransform(BasicTransform):
    def __init__(self, transform, apply_probability):
        self.transform, self.apply_probability = transform, apply_probability

    def bind(self, scratch):
        self.transform.bind(scratch)
        return super().bind(scratch)

    def __call__(self, image):
        if np.random.uniform() < self.apply_probability:
            self.transform(image)


class SpatialTransform(BasicTransform):
    """Rotation / scaling about the patch centre, trilinear, zeros outside (elastic deformation and random cropping are switched off
    by the reference, DATA:766-773).  Without a draw the patch is returned as it is (patch_size == the cut patch)."""

    def __init__(self, patch_size, p_rotation=0, rotation=None, p_scaling=0, scaling=None, p_synchronize_scaling_across_axes=None, **unused):
        self.patch_size = tuple(patch_size)
        self.p_rotation, self.rotation, self.p_scaling, self.scaling = p_rotation, rotation, p_scaling, scaling
        self.p_sync = p_synchronize_scaling_across_axes or 0

    def draw(self, image):
        nd = image.dim() - 1
        do_rot, do_scale = np.random.uniform() < self.p_rotation, np.random.uniform() < self.p_scaling
        if not (do_rot or do_scale):
            return None
        if do_rot:  # 3-D: one angle per axis; 2-D: the single in-plane angle is asked for with dim=0
            angles = [sample_scalar(self.rotation, image=image, dim=i) for i in range(3 if nd == 3 else 1)]
            angles = angles if nd == 3 else [angles[0], 0.0, 0.0]  # a 2-D plane is [1, H, W]: in-plane = about axis 0
        else:
            angles = [0.0, 0.0, 0.0]
        if do_scale:
            if np.random.uniform() < self.p_sync:
                scales = [sample_scalar(self.scaling, image=image, dim=None)] * 3
            else:
                scales = [sample_scalar(self.scaling, image=image, dim=i) for i in range(nd)]
                scales = scales if nd == 3 else [1.0] + scales
        else:
            scales = [1.0, 1.0, 1.0]
        return affine_matrix(angles, scales)

    def __call__(self, image):
        if tuple(image.shape[1:]) != self.patch_size:
            raise ValueError(f"SpatialTransform(patch_size={self.patch_size}) on a patch of {tuple(image.shape[1:])}: the loader cuts the "
                             "final patch size (initial_patch_size == patch_size in the soft setting, DATA:409)")
        m = self.draw(image)
        if m is None:
            return
        for c in range(image.shape[0]):  # one matrix for all channels of the sample
            tmp = self.scratch.plane(0, image[c])
            tmp.copy_(image[c])
            affine_sample(tmp, image[c], m)


class GaussianNoiseTransform(BasicTransform):
    def __init__(self, noise_variance=(0, 0.1), p_per_channel=1, synchronize_channels=True):
        self.noise_variance, self.p_per_channel, self.synchronize_channels = noise_variance, p_per_channel, synchronize_channels

    def __call__(self, image):
        chans = [c for c in range(image.shape[0]) if np.random.uniform() < self.p_per_channel]
        if not chans:
            return
        if self.synchronize_channels:  # one sigma and ONE noise field for every channel (the value is used as the std, as upstream does)
            sigma = sample_scalar(self.noise_variance, image=image)
            field = torch.randn(image.shape[1:], dtype=F32, device=image.device)
            for c in chans:
                pointwise(image[c], ADD_NOISE, sigma, aux=field)
        else:
            for c in chans:
                pointwise(image[c], ADD_NOISE, sample_scalar(self.noise_variance, image=image),
                          aux=torch.randn(image.shape[1:], dtype=F32, device=image.device))


class GaussianBlurTransform(BasicTransform):
    def __init__(self, blur_sigma=(0.5, 1.0), synchronize_channels=False, synchronize_axes=False, p_per_channel=0.5, benchmark=False):
        self.blur_sigma, self.sync_c, self.sync_a, self.p_per_channel = blur_sigma, synchronize_channels, synchronize_axes, p_per_channel

    def __call__(self, image):
        nd = image.dim() - 1
        shared = None
        for c in range(image.shape[0]):
            if np.random.uniform() >= self.p_per_channel:
                continue
            if self.sync_c and shared is not None:
                sigmas = shared
            else:
                sigmas = [sample_scalar(self.blur_sigma, image=image)] * nd if self.sync_a else [sample_scalar(self.blur_sigma, image=image)
                                                                                                  for _ in range(nd)]
                shared = sigmas
            a, b = self.scratch.plane(0, image[c]), self.scratch.plane(1, image[c])
            hops = [image[c], a, b, image[c]] if nd == 3 else [image[c], a, image[c]]
            for ax in range(nd):
                blur_axis(hops[ax], hops[ax + 1], ax, gaussian_taps(sigmas[ax]))


class MultiplicativeBrightnessTransform(BasicTransform):
    def __init__(self, multiplier_range, synchronize_channels=False, p_per_channel=1):
        self.multiplier_range, self.sync_c, self.p_per_channel = multiplier_range, synchronize_channels, p_per_channel

    def __call__(self, image):
        m = sample_scalar(self.multiplier_range, image=image) if self.sync_c else None
        for c in range(image.shape[0]):
            if np.random.uniform() < self.p_per_channel:
                pointwise(image[c], SCALE, m if m is not None else sample_scalar(self.multiplier_range, image=image))


class ContrastTransform(BasicTransform):
    def __init__(self, contrast_range, preserve_range=True, synchronize_channels=False, p_per_channel=1):
        if not preserve_range:
            raise NotImplementedError("ContrastTransform(preserve_range=False): the reference sets True (DATA:802)")
        self.contrast_range, self.sync_c, self.p_per_channel = contrast_range, synchronize_channels, p_per_channel

    def __call__(self, image):
        f = sample_scalar(self.contrast_range, image=image) if self.sync_c else None
        s = self.scratch
        for c in range(image.shape[0]):
            if np.random.uniform() < self.p_per_channel:
                plane_stats(image[c], s.stats[0], s.work)
                pointwise(image[c], CONTRAST, f if f is not None else sample_scalar(self.contrast_range, image=image), stats_a=s.stats[0])


class SimulateLowResolutionTransform(BasicTransform):
    def __init__(self, scale=(0.5, 1), synchronize_channels=False, synchronize_axes=True, ignore_axes=None, allowed_channels=None,
                 p_per_channel=0.5):
        if not synchronize_axes or ignore_axes:
            raise NotImplementedError("per-axis / ignored-axis low-resolution scales: the reference sets synchronize_axes=True (DATA:811)")
        self.scale, self.sync_c, self.allowed, self.p_per_channel = scale, synchronize_channels, allowed_channels, p_per_channel

    def __call__(self, image):
        shared = sample_scalar(self.scale, image=image) if self.sync_c else None
        for c in (range(image.shape[0]) if self.allowed is None else self.allowed):
            if np.random.uniform() >= self.p_per_channel:
                continue
            sc = shared if shared is not None else sample_scalar(self.scale, image=image)
            low = [max(1, round(n * sc)) for n in image.shape[1:]]
            tmp = self.scratch.plane(0, image[c])
            tmp.copy_(image[c])
            lowres(tmp, image[c], low)


class GammaTransform(BasicTransform):
    def __init__(self, gamma, p_invert_image=0, synchronize_channels=False, p_per_channel=1, p_retain_stats=1):
        self.gamma, self.p_invert, self.sync_c, self.p_per_channel, self.p_retain = gamma, p_invert_image, synchronize_channels, p_per_channel, p_retain_stats

    def __call__(self, image):
        s = self.scratch
        g_shared = sample_scalar(self.gamma, image=image) if self.sync_c else None
        for c in range(image.shape[0]):
            if np.random.uniform() >= self.p_per_channel:
                continue
            invert, retain = np.random.uniform() < self.p_invert, np.random.uniform() < self.p_retain
            g = g_shared if g_shared is not None else sample_scalar(self.gamma, image=image)
            if invert:
                pointwise(image[c], SCALE, -1.0)
            plane_stats(image[c], s.stats[1], s.work)  # "before": min / range of the gamma curve, mean / std to restore
            pointwise(image[c], GAMMA, g, stats_a=s.stats[1])
            if retain:
                plane_stats(image[c], s.stats[0], s.work)
                pointwise(image[c], RESTORE_STATS, stats_a=s.stats[0], stats_b=s.stats[1])
            if invert:
                pointwise(image[c], SCALE, -1.0)


class MirrorTransform(BasicTransform):
    def __init__(self, allowed_axes):
        self.allowed_axes = tuple(allowed_axes)

    def __call__(self, image):
        nd = image.dim() - 1
        mask = 0
        for ax in self.allowed_axes:
            if np.random.uniform() < 0.5:
                mask |= 1 << (ax + 3 - nd)  # bit 0 / 1 / 2 = D / H / W of mi_crop_pad
        if mask:
            from .data import crop_and_pad_nd
            tmp = self.scratch.plane(("img", image.shape[0]), image)
            tmp.copy_(image)
            crop_and_pad_nd(tmp, [[0, n] for n in image.shape[1:]], 0, flip_mask=mask >> (3 - nd), out=image)


class ComposeTransforms:
    def __init__(self, transforms):
        self.transforms = list(transforms)
        self.scratch = _Scratch()
        for t in self.transforms:
            t.bind(self.scratch)

    def _one(self, image):
        for c in range(image.shape[0]):
            _check(image[c])
        self.scratch.bind(image.device)
        for t in self.transforms:
            t(image)
        return image

    def __call__(self, batch=None, *, image=None):
        """chain(image=t[C, *spatial]) -> {'image': t} like the reference's call (DATA:531); chain(batch[B, C, *spatial]) -> batch for
        GpuPatchLoader(transform=).  Works in place on a contiguous fp32 device tensor."""
        if image is not None:
            return {"image": self._one(image)}
        for b in range(batch.shape[0]):
            self._one(batch[b])
        return batch


def define_nnunet_transformations(params, validation=False):
    """DATA:748-859 with the same `params` keys; `validation=True`: the centre-crop-only SpatialTransform, i.e. nothing to do on a
    patch that already has `patch_size`."""
    transforms = []
    if validation:
        return ComposeTransforms([SpatialTransform(params["patch_size"], p_rotation=0, p_scaling=0)])
    if params.get("dummy_2d"):
        raise NotImplementedError("dummy_2d (Convert3DTo2DTransform): MedicalDataset's soft setting never sets it (DATA:408)")
    transforms.append(SpatialTransform(
        params["patch_size"], p_rotation=0.2 if params["rotation"] else 0, rotation=params["rot_for_da"] if params["rotation"] else None,
        p_scaling=0.2 if params["scaling"] else 0, scaling=params["scaling_range"] if params["scaling"] else None,
        p_synchronize_scaling_across_axes=1 if params["scaling"] else None))
    if params["gaussian_noise"]:
        transforms.append(RandomTransform(GaussianNoiseTransform(noise_variance=(0, 0.1), p_per_channel=1, synchronize_channels=True), 0.1))
    if params["gaussian_blur"]:
        transforms.append(RandomTransform(GaussianBlurTransform(blur_sigma=(0.5, 1.0), synchronize_channels=False, synchronize_axes=False,
                                                                p_per_channel=0.5), 0.2))
    if params["brightness"]:
        transforms.append(RandomTransform(MultiplicativeBrightnessTransform(BGContrast(params["brightness_range"]), False, 1), 0.15))
    if params["contrast"]:
        transforms.append(RandomTransform(ContrastTransform(BGContrast(params["contrast_range"]), True, False, 1), 0.15))
    if params["low_resolution"]:
        transforms.append(RandomTransform(SimulateLowResolutionTransform(scale=(0.5, 1), synchronize_channels=False, synchronize_axes=True,
                                                                         ignore_axes=None, allowed_channels=None, p_per_channel=0.5), 0.25))
    if params["gamma"]:
        g = BGContrast(params["gamma_range"])
        transforms.append(RandomTransform(GammaTransform(g, p_invert_image=1, p_per_channel=1, p_retain_stats=1), 0.0))  # DATA:819-828: never fires
        transforms.append(RandomTransform(GammaTransform(g, p_invert_image=0, p_per_channel=1, p_retain_stats=1), 0.3))
    if params.get("mirror_axes"):
        transforms.append(MirrorTransform(params["mirror_axes"]))
    return ComposeTransforms(transforms)


def soft_setting(transformation_args, dim=3):
    """MedicalDataset.__init__ / configure_augmentation_params(heavy_augmentation=False) (DATA:288-295, 399-424): fills the derived
    keys of the user's `transformation_args` (switches rotation / scaling / mirror / brightness / contrast / gamma / dummy_2d +
    patch_size) for the soft augmentation the generative trainers use: rotation about the depth axis by at most 10 degrees, scaling /
    brightness / contrast / gamma in (0.9, 1.1), mirroring along the last axis."""
    rot_dim = 0 if dim == 3 else 2  # DATA:407 (with 2-D data no axis ever matches: the reference's 2-D patches are not rotated)

    def rot(image=None, dim=None):
        return np.random.uniform(-0.174533, 0.174533) if dim == rot_dim else 0.0

    a = dict(transformation_args)
    a["rot_for_da"] = rot if a.get("rotation") else None
    a["dummy_2d"] = False if a.get("dummy_2d") else None  # do_dummy_2d is False in the soft setting
    a["mirror_axes"] = ((2,) if dim == 3 else (1,)) if a.get("mirror") else None
    a["scaling_range"] = (0.9, 1.1) if a.get("scaling") else None
    a["brightness_range"] = (0.9, 1.1) if a.get("brightness") else None
    a["contrast_range"] = (0.9, 1.1) if a.get("contrast") else None
    a["gamma_range"] = (0.9, 1.1) if a.get("gamma") else None
    for k in ("rotation", "scaling", "gaussian_noise", "gaussian_blur", "brightness", "contrast", "low_resolution", "gamma"):
        a.setdefault(k, False)
    return a
