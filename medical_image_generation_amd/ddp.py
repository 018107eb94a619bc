"""Data-parallel gradient exchange for the flat gradient arena.

The reference is single-process; this is the one exchange step a data-parallel train step adds: average the trainable
prefix of the flat fp32 gradient buffer over the ranks.  One process per GPU, `torch.distributed` backend "nccl" (= RCCL
over xGMI on MI355X).  Because the gradients already live in ONE contiguous buffer there is no per-parameter bucketing
logic: the buffer is cut into a few large slices (default 64 MiB: large enough to run each xGMI link at its streaming
rate, small enough that the first slice's all-reduce overlaps the launch of the next), pre-scaled by 1/world so the SUM
all-reduce yields the mean with no extra pass.  Statically unused parameters (`proj_attn.*`) sit behind the trainable
prefix and are never communicated.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_slices(numel: int, bucket_elems: int):
    """[(start, end)] covering [0, numel) in order."""
    if bucket_elems <= 0:
        raise ValueError("bucket_elems must be positive")
    return [(o, min(o + bucket_elems, numel)) for o in range(0, numel, bucket_elems)]


def average_gradients(flat_grad: torch.Tensor, n_trainable: int, group=None, bucket_elems: int = 16 << 20, async_op: bool = False):
    """In-place mean over ranks of flat_grad[:n_trainable].  Returns the list of work handles when async_op."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world <= 1:
        return []
    g = flat_grad[:n_trainable]
    g.mul_(1.0 / world)
    works = []
    for a, b in bucket_slices(n_trainable, bucket_elems):
        w = dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def broadcast_parameters(flat_data: torch.Tensor, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters (whole arena, including the unused tail)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)
