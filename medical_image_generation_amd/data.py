"""Data path on the MI355X: the dataset lives in HBM, patches are cut on the GPU.

The reference (medimgen/data_processing.py, DATA) opens a zarr / blosc2 / npy volume per sample in a DataLoader worker, crops and
pads a patch on the CPU (`crop_and_pad_nd`, DATA:148-225; box from `MedicalDataset.get_bbox`, DATA:473-528), runs
batchgeneratorsv2 transforms, clamps to [0, 1] (DATA:595) and ships the batch over PCIe.  With 288 GB of HBM3E per GPU the whole
training set of a medical-imaging task fits next to the model (a 512^3 fp16 volume is 268 MB), so here the volumes are uploaded ONCE
(`ResidentDataset`, fp16 or fp32) and every batch is cut by HIP kernels straight into the fp32 NC[D]HW tensor the trainers read:
no host copy, no worker processes, nothing on PCIe in the step.

Mirrors of the reference (same names, argument meaning and RNG call order on numpy's global generator, so a seeded run draws the same
boxes the reference's loader would):
  crop_and_pad_nd(image, bbox, pad_value)   -- torch GPU tensors: one launch of `mi_crop_pad`
  PatchSampler.get_bbox / oversampling      -- DATA:426-428, 473-528 (host logic: a few integers per sample)
  BatchOrder                                -- CustomBatchSampler, DATA:601-643 (every sample once before any repeats; 250 steps/epoch)
  GpuPatchLoader                            -- __getitem__ + collate: {'id': [...], 'image': [B, C, *patch] fp32 in [0, 1]}
Augmentation: the mirror of the soft setting (DATA:410) and a multiplicative brightness draw ride in the same kernel; the rest of
the reference's transform list is third-party batchgeneratorsv2 code whose semantics nothing under /root/reference pins -- it is not
restated here (PARITY UNPINNED, see oracle/data.py), and `transform=` takes any callable on the GPU batch instead.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import call, ptr

F32 = torch.float32


def crop_and_pad_nd(image: torch.Tensor, bbox, pad_value=0, flip_mask: int = 0, scale: float = 1.0, clamp01: bool = False,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    """DATA:148-225 for a GPU tensor: crop `bbox` = [[lo, hi), ...] from the last len(bbox) (<= 3) axes, pad what lies outside
    with pad_value; a box completely outside the image gives zeros (DATA:191-197).  fp32 or fp16 source, fp32 result."""
    if not image.is_cuda:
        raise RuntimeError("medical_image_generation_amd.data cuts patches on the GPU: move the volume to 'cuda' (ResidentDataset does)")
    if image.dtype not in (torch.float32, torch.float16) or not image.is_contiguous():
        raise ValueError("volume must be a contiguous fp32 or fp16 tensor")
    k = len(bbox)
    if not 1 <= k <= 3 or k > image.dim():
        raise ValueError("bbox must cover the last 1-3 axes")
    lead = image.shape[:image.dim() - k]
    sp = (1,) * (3 - k) + tuple(image.shape[image.dim() - k:])
    lo = [0] * (3 - k) + [int(b[0]) for b in bbox]
    size = [1] * (3 - k) + [int(b[1]) - int(b[0]) for b in bbox]
    if any(s <= 0 for s in size):
        raise ValueError("empty bounding box")
    target = tuple(lead) + tuple(size[3 - k:])
    if out is None:
        out = torch.empty(target, dtype=F32, device=image.device)
    elif tuple(out.shape) != target or out.dtype != F32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 tensor of the target shape")
    if any(l + s <= 0 or l >= n for l, s, n in zip(lo, size, sp)):
        return out.zero_()
    c = 1
    for v in lead:
        c *= int(v)
    call("mi_crop_pad", ptr(image), int(image.dtype == torch.float16), c, sp[0], sp[1], sp[2], (C.c_int * 3)(*lo), ptr(out), size[0], size[1],
         size[2], float(pad_value), int(flip_mask) << (3 - k) if k < 3 else int(flip_mask), float(scale), int(clamp01))
    return out


class ResidentDataset:
    """The training volumes, uploaded once and kept in HBM.  add(name, volume [C, D, H, W] or [C, H, W] numpy / torch, class_locations):
    `class_locations` = {class: array of (c, z, y, x) voxel coordinates} as the reference's preprocessing stores it in the .pkl
    (DATA:556-557), used for foreground oversampling."""

    def __init__(self, device="cuda", dtype=torch.float16):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError("dtype must be torch.float16 or torch.float32")
        self.device, self.dtype = torch.device(device), dtype
        self.ids, self.volumes, self.class_locations = [], {}, {}

    def add(self, name, volume, class_locations=None):
        v = torch.as_tensor(np.ascontiguousarray(volume) if isinstance(volume, np.ndarray) else volume)
        self.volumes[name] = v.to(self.device, self.dtype).contiguous()
        self.class_locations[name] = class_locations
        self.ids.append(name)

    def add_case(self, data_path, name):
        """One preprocessed case from disk, as MedicalDataset.load_image finds it (DATA:535-561: <name>.zarr, .npy or .npz, properties
        in <name>.pkl): read once on the host (volume_io) and uploaded."""
        from .volume_io import load_image
        image, properties = load_image(data_path, name)
        self.add(name, np.asarray(image), properties.get("class_locations"))

    def __len__(self):
        return len(self.ids)

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.volumes.values())


class PatchSampler:
    """The box logic of MedicalDataset (soft augmentation: initial_patch_size == patch_size unless given)."""

    def __init__(self, patch_size, batch_size, oversample_foreground_percent=0.0, initial_patch_size=None, probabilistic_oversampling=False):
        patch_size = tuple(int(p) for p in patch_size)
        self.is_2d = len(patch_size) == 2
        self.patch_size = (1,) + patch_size if self.is_2d else patch_size  # pseudo 3-D (DATA:298-299)
        ips = tuple(int(p) for p in (initial_patch_size or patch_size))
        self.initial_patch_size = (1,) + ips if len(ips) == 2 else ips
        self.need_to_pad = (np.array(self.initial_patch_size) - np.array(self.patch_size)).astype(int)
        self.batch_size, self.oversample_foreground_percent = batch_size, oversample_foreground_percent
        self.probabilistic = probabilistic_oversampling

    def force_foreground(self, batch_idx: int) -> bool:
        if self.probabilistic:  # DATA:431-433
            return np.random.uniform() < self.oversample_foreground_percent
        return batch_idx >= round(self.batch_size * (1 - self.oversample_foreground_percent))  # DATA:426-428

    # ---- box placement, one axis at a time.  What the reference's loader does (MedicalDataset.get_bbox, DATA:473-528), stated as
    # three per-axis rules; the draws on numpy's global generator come in the reference's order (one uniform integer per axis, then --
    # foreground only -- a class and a voxel, then one jitter per in-plane axis that has room) so a seeded run cuts the same boxes.
    def _span(self, axis, extent):
        """[first, last] admissible lower corner along `axis` of a volume of `extent` voxels.  The box may overhang by the slack the
        (initial) patch has over the final one, split low/high with the odd voxel high; a volume shorter than the box gets exactly
        the overhang that centres it."""
        box = self.initial_patch_size[axis]
        slack = max(int(self.need_to_pad[axis]), box - extent)
        return -slack // 2, extent + slack // 2 + slack % 2 - box

    def _towards(self, axis, extent, voxel_coord):
        """Lower corner that centres the box on a foreground voxel, kept inside _span."""
        first, last = self._span(axis, extent)
        return max(first, min(int(voxel_coord) - self.initial_patch_size[axis] // 2, last))

    def _in_plane(self, axis, extent):
        """H / W: the box sits on the volume's centre, shifted by a uniform jitter of at most 10 voxels (less when the margin on
        either side is smaller; none -- and no draw -- when there is no margin or the volume is narrower than the box)."""
        box, mid = self.initial_patch_size[axis], extent // 2
        if extent >= box:
            room = min(10, mid - box // 2, extent - mid - (box - box // 2))
            if room > 0:
                mid += np.random.randint(-room, room + 1)
        return mid - box // 2

    def get_bbox(self, data_shape, force_fg, class_locations):
        """(lower corners, upper corners) of the next patch in a volume of spatial shape `data_shape` (DATA:473-528)."""
        axes = range(len(data_shape))
        lower = []
        for ax in axes:  # every axis draws, also those overwritten below: the generator's state must advance as the reference's does
            first, last = self._span(ax, data_shape[ax])
            lower.append(np.random.randint(first, last + 1))
        classes = [c for c in class_locations if len(class_locations[c]) > 0] if (force_fg and class_locations is not None) else []
        if classes:
            voxels = class_locations[np.random.choice(classes)]
            voxel = voxels[np.random.choice(len(voxels))]
            if self.is_2d:
                lower[0] = voxel[0]  # pseudo 3-D: the slice that holds the voxel
            else:
                lower = [self._towards(ax, data_shape[ax], voxel[ax]) for ax in axes]
        for ax in list(axes)[-2:]:
            lower[ax] = self._in_plane(ax, data_shape[ax])
        return lower, [lo + self.initial_patch_size[ax] for ax, lo in zip(axes, lower)]


class BatchOrder:
    """The epoch schedule of the reference's CustomBatchSampler (DATA:601-643): `number_of_steps` batches per epoch; within an epoch
    every sample is dealt once before any is dealt again (a new shuffled deck whenever fewer than a batch of cards is left).  The deck
    is a member and is shuffled IN PLACE at the start of every epoch, so epoch k starts from epoch k-1's permutation like the
    reference's `self.indices` does (a deck rebuilt from range(n) each epoch would draw other batches from epoch 2 on)."""

    def __init__(self, n_items, batch_size, number_of_steps=250, shuffle=True):
        self.n_items, self.batch_size, self.number_of_steps, self.shuffle = n_items, batch_size, number_of_steps, shuffle
        self.indices = list(range(n_items))

    def __len__(self):
        return self.number_of_steps

    def _deal(self):
        """Sample indices of one epoch as one flat run, `number_of_steps * batch_size` long."""
        if self.shuffle:
            np.random.shuffle(self.indices)
        want = self.number_of_steps * self.batch_size
        deck, run = list(self.indices), []
        while len(run) < want:
            if len(deck) < self.batch_size:  # fewer cards than a batch: the leftovers go away with the old deck
                deck = list(self.indices)
                if self.shuffle:
                    np.random.shuffle(deck)
            run += deck[:self.batch_size]
            del deck[:self.batch_size]
        return run

    def __iter__(self):
        run, b = self._deal(), self.batch_size
        for k in range(self.number_of_steps):
            yield list(enumerate(run[k * b:(k + 1) * b]))


class GpuPatchLoader:
    """for batch in loader: batch['image'] is a fresh fp32 [B, C, *patch] tensor in [0, 1] cut from the resident volumes (2-D patch
    sizes drop the slice axis, DATA:584), batch['id'] the sample names.  section='training': shuffled order, optional mirror along the
    soft setting's axis (last axis; each with probability 0.5) and brightness multiplier in `brightness_range` with probability 0.15
    (DATA:410-413, 790-797); anything else: validation (no shuffle draw beyond the reference's, no augmentation)."""

    def __init__(self, dataset: ResidentDataset, patch_size, batch_size, number_of_steps=250, section="training", oversample_foreground_percent=0.0,
                 mirror=False, brightness_range=None, channel_ids=None, transform=None):
        self.ds, self.batch_size, self.section = dataset, batch_size, section
        self.sampler = PatchSampler(patch_size, batch_size, oversample_foreground_percent)
        self.order = BatchOrder(len(dataset), batch_size, number_of_steps, shuffle=section == "training")
        self.mirror, self.brightness_range, self.channel_ids, self.transform = mirror, brightness_range, channel_ids, transform
        self._selected = {}  # id -> the channel-selected copy of its resident volume (made once, not per sample)

    def __len__(self):
        return len(self.order)

    def __iter__(self):
        s = self.sampler
        for batch in self.order:
            first = self.ds.volumes[self.ds.ids[batch[0][1]]]
            chans = first.shape[0] if self.channel_ids is None else len(self.channel_ids)
            out = torch.empty((len(batch), chans) + s.patch_size, dtype=F32, device=first.device)
            names = []
            for batch_idx, sample_idx in batch:
                name = self.ds.ids[sample_idx]
                vol = self.ds.volumes[name]
                vol4 = vol if vol.dim() == 4 else vol.unsqueeze(1)  # [C, D, H, W] (2-D data: one slice)
                lbs, ubs = s.get_bbox(tuple(vol4.shape[1:]), s.force_foreground(batch_idx), self.ds.class_locations[name])
                flip, scale = 0, 1.0
                if self.section == "training":
                    if self.mirror and np.random.uniform() < 0.5:
                        flip = 4  # W axis: mirror_axes = (2,) in 3-D / (1,) in 2-D (DATA:410)
                    if self.brightness_range is not None and np.random.uniform() < 0.15:
                        scale = float(np.random.uniform(*self.brightness_range))
                src = vol4
                if self.channel_ids is not None:
                    src = self._selected.get(name)
                    if src is None:
                        src = self._selected[name] = vol4[list(self.channel_ids)].contiguous()
                crop_and_pad_nd(src, [[a, b] for a, b in zip(lbs, ubs)], 0, flip_mask=flip, scale=scale, clamp01=True, out=out[batch_idx])
                names.append(name)
            image = out.squeeze(2) if s.is_2d else out
            if self.transform is not None:  # the reference clamps AFTER its transform list (DATA:590-595)
                image = self.transform(image).clamp_(0, 1)
            yield {"id": names, "image": image}
