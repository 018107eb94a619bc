"""Oracle for the data path -- TEST INFRASTRUCTURE (see oracle/__init__.py).

numpy restatements of the pure-Python / numpy pieces of medimgen/data_processing.py (DATA):
  crop_and_pad_nd            DATA:148-225
  MedicalDataset.get_bbox    DATA:473-528 (+ need_to_pad, DATA:302)
  _oversample_last_XX_percent DATA:426-428
  CustomBatchSampler         DATA:601-643

PARITY UNPINNED for everything here: data_processing.py imports zarr, blosc2 and batchgeneratorsv2 at module scope, none of which
is installed or installable in this image, so the reference functions cannot be executed and the reference holds no vectors for
them.  The restatements follow the source line by line in behaviour (same RNG calls in the same order on numpy's global generator,
so a seeded run draws the same boxes the reference's loader would) and are cross-checked in tests/test_data_cpu.py against an
independent formulation (pad-the-whole-volume-then-slice).
"""
from __future__ import annotations

import numpy as np


def crop_and_pad_nd(image: np.ndarray, bbox, pad_value=0) -> np.ndarray:
    """Crop `bbox` = [[lo, hi), ...] from the LAST len(bbox) axes (upper bound excluded); whatever lies outside the image is
    pad_value; a box completely outside along any axis gives an all-ZERO result of the target shape (DATA:191-197: zeros, not
    pad_value)."""
    k = len(bbox)
    lead = image.ndim - k
    target = list(image.shape[:lead]) + [hi - lo for lo, hi in bbox]
    slices, pads = [slice(None)] * lead, [(0, 0)] * lead
    for a, (lo, hi) in enumerate(bbox):
        n = image.shape[lead + a]
        if hi <= 0 or lo >= n:
            return np.zeros(target, dtype=image.dtype)
        slices.append(slice(max(lo, 0), min(hi, n)))
        pads.append((max(0, -lo), max(0, hi - n)))
    return np.pad(image[tuple(slices)], pads, mode="constant", constant_values=pad_value)


class BBoxSampler:
    """get_bbox of MedicalDataset for the soft-augmentation setting (initial_patch_size == patch_size, DATA:410)."""

    def __init__(self, patch_size, initial_patch_size=None):
        self.patch_size = tuple(patch_size)
        self.initial_patch_size = tuple(initial_patch_size or patch_size)
        self.need_to_pad = (np.array(self.initial_patch_size) - np.array(self.patch_size)).astype(int)

    def get_bbox(self, data_shape, force_fg, class_locations, is_2d=False):
        dim = len(data_shape)
        need_to_pad = self.need_to_pad.copy()
        for d in range(dim):
            if need_to_pad[d] + data_shape[d] < self.initial_patch_size[d]:
                need_to_pad[d] = self.initial_patch_size[d] - data_shape[d]
        lbs = [-need_to_pad[i] // 2 for i in range(dim)]
        ubs = [data_shape[i] + need_to_pad[i] // 2 + need_to_pad[i] % 2 - self.initial_patch_size[i] for i in range(dim)]
        bbox_lbs = [np.random.randint(lbs[i], ubs[i] + 1) for i in range(dim)]
        if force_fg and class_locations is not None:
            eligible = [cls for cls in class_locations if len(class_locations[cls]) > 0]
            if eligible:
                selected_class = np.random.choice(eligible)
                voxels = class_locations[selected_class]
                selected_voxel = voxels[np.random.choice(len(voxels))]
                for i in range(dim):
                    if is_2d and i == 0:
                        bbox_lbs[0] = selected_voxel[0]
                    elif not is_2d:
                        bbox_lbs[i] = max(lbs[i], min(selected_voxel[i] - self.initial_patch_size[i] // 2, ubs[i]))
        for i in range(dim - 2, dim):  # H and W: centre crop with a jitter of at most 10 voxels (DATA:509-523)
            crop, size = self.initial_patch_size[i], data_shape[i]
            center = size // 2
            if size < crop:
                bbox_lbs[i] = center - crop // 2
            else:
                max_offset = min(10, center - crop // 2, size - center - (crop - crop // 2))
                offset = np.random.randint(-max_offset, max_offset + 1) if max_offset > 0 else 0
                bbox_lbs[i] = center + offset - crop // 2
        return bbox_lbs, [bbox_lbs[i] + self.initial_patch_size[i] for i in range(dim)]


def oversample_last_percent(sample_idx: int, batch_size: int, oversample_foreground_percent: float) -> bool:
    return sample_idx >= round(batch_size * (1 - oversample_foreground_percent))


def batch_sample_order(n_items: int, batch_size: int, number_of_steps: int = 250, shuffle: bool = True):
    """CustomBatchSampler.define_indices + __iter__: list of batches of (position in batch, sample index)."""
    indices = list(range(n_items))
    if shuffle:
        np.random.shuffle(indices)
    order, available = [], indices.copy()
    while len(order) < number_of_steps * batch_size:
        if len(available) < batch_size:
            available = indices.copy()
            if shuffle:
                np.random.shuffle(available)
        order.extend(available[:batch_size])
        available = available[batch_size:]
    return [[(i, s) for i, s in enumerate(order[k * batch_size:(k + 1) * batch_size])] for k in range(number_of_steps)]


# ---------------------------------------------------------------------------------------------------------------------------------
# The transform list of define_nnunet_transformations (DATA:748-859).  The classes are batchgeneratorsv2's (third-party, absent from
# /root/reference and from this image): PARITY UNPINNED.  Each function states ONE transform's arithmetic for given parameters with
# torch's own CPU ops (F.pad / conv, F.interpolate, F.grid_sample), which is what that package calls; the draws are the caller's.
def _torch():
    import torch
    import torch.nn.functional as F
    return torch, F


def aug_contrast(plane, factor):
    """ContrastTransform(preserve_range=True) on one channel."""
    mean, lo, hi = plane.mean(), plane.min(), plane.max()
    return ((plane - mean) * factor + mean).clamp(lo, hi)


def aug_gamma(plane, gamma, invert=False, retain_stats=True):
    """GammaTransform on one channel."""
    torch, _ = _torch()
    x = -plane if invert else plane.clone()
    mean, std = x.mean(), x.std()
    lo = x.min()
    rng = x.max() - lo
    x = torch.pow((x - lo) / rng.clamp(min=1e-7), gamma) * rng + lo
    if retain_stats:
        x = (x - x.mean()) * (std / x.std().clamp(min=1e-7)) + mean
    return -x if invert else x


def aug_gaussian_taps(sigma, truncate=6.0):
    torch, _ = _torch()
    k = int(round(sigma * truncate + 0.5))
    k += 1 - k % 2
    ax = torch.arange(k, dtype=torch.float32) - k // 2
    w = torch.exp(-0.5 * (ax / sigma) ** 2)
    return w / w.sum()


def aug_blur(plane, sigmas):
    """GaussianBlurTransform on one channel: separable, one sigma per spatial axis, reflect padding."""
    torch, F = _torch()
    x = plane.clone()
    nd = x.dim()
    for ax, sigma in enumerate(sigmas):
        w = aug_gaussian_taps(sigma)
        r = len(w) // 2
        xm = x.movedim(ax, -1)
        shp = xm.shape
        rows = F.pad(xm.reshape(-1, 1, shp[-1]), (r, r), mode="reflect")
        x = F.conv1d(rows, w.view(1, 1, -1)).reshape(shp).movedim(-1, ax)
    assert x.dim() == nd
    return x


def aug_lowres(plane, low_shape):
    """SimulateLowResolutionTransform on one channel: nearest-exact down to low_shape, (bi/tri)linear back."""
    _, F = _torch()
    mode = {2: "bilinear", 3: "trilinear"}[plane.dim()]
    down = F.interpolate(plane[None, None], size=tuple(low_shape), mode="nearest-exact")
    return F.interpolate(down, size=tuple(plane.shape), mode=mode)[0, 0]


def aug_affine(plane, matrix, out_shape=None):
    """SpatialTransform without deformation on one channel: centred output grid -> source = matrix @ g + centre, sampled with
    grid_sample(bilinear, zeros, align_corners=False).  `matrix` acts on (d, h, w) (2-D: embed as [1, H, W])."""
    torch, F = _torch()
    x = plane if plane.dim() == 3 else plane[None]
    out_shape = tuple(out_shape or x.shape)
    m = torch.as_tensor(matrix, dtype=torch.float32)
    axes = [torch.arange(n, dtype=torch.float32) - (n - 1) / 2 for n in out_shape]
    g = torch.stack(torch.meshgrid(*axes, indexing="ij"), -1)  # [..., 3] centred (d, h, w)
    src = g @ m.T + torch.tensor([(n - 1) / 2 for n in x.shape])
    # voxel index p <-> normalised coordinate (2 p + 1) / n - 1 under align_corners=False; grid_sample wants (w, h, d) order
    norm = (2 * src + 1) / torch.tensor([float(n) for n in x.shape]) - 1
    y = F.grid_sample(x[None, None], norm.flip(-1)[None], mode="bilinear", padding_mode="zeros", align_corners=False)[0, 0]
    return y if plane.dim() == 3 else y[0]
