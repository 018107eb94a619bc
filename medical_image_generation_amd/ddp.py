"""Data-parallel gradient exchange for the flat gradient arena.

The reference is single-process; this is the one exchange step a data-parallel train step adds: combine the trainable prefix of
the flat fp32 gradient buffer over the ranks.  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on
MI355X; "gloo" on CPU for the tests).

* The gradients already live in ONE contiguous buffer, so there is no per-parameter bucketing logic: a segment is cut into a few
  large slices (default 64 MiB: large enough to run each xGMI link at its streaming rate; xGMI is point-to-point, 7 links per
  GPU, so a collective is per-link bound and gains nothing from many small messages).
* The arena is ordered by gradient-completion time (engine.ParamArena): the *early* segment [n_late, n_trainable) is final when
  the backward pass reaches the model's cut mark; its SUM all-reduce is issued there, asynchronously, and overlaps the rest of the
  backward; the small *late* prefix [0, n_late) follows when the backward is done (`GradientExchange`).
* SUM, not mean: 1/world is folded into the optimizer kernel (`mi_adam_step(grad_scale)`), so no pass over the buffer scales it.
* Statically unused parameters (`proj_attn.*`) sit behind the trainable prefix and are never communicated.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_slices(numel: int, bucket_elems: int, start: int = 0):
    """[(a, b)] covering [start, start + numel) in order."""
    if bucket_elems <= 0:
        raise ValueError("bucket_elems must be positive")
    return [(start + o, start + min(o + bucket_elems, numel)) for o in range(0, numel, bucket_elems)]


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def sum_gradients(flat_grad: torch.Tensor, lo: int, hi: int, group=None, bucket_elems: int = 16 << 20, async_op: bool = False,
                  force: bool = False):
    """In-place SUM over ranks of flat_grad[lo:hi].  Returns the work handles when async_op (else []).  force: issue the collectives
    even in a one-rank group (a sum over one rank changes nothing, but the collective library's launch, stream and event path runs:
    how the RCCL calls between the hipGraph replays are rehearsed on a single GPU, tests/test_trainer_gpu.py)."""
    if hi <= lo or (_world(group) <= 1 and not (force and dist.is_initialized())):
        return []
    works = []
    for a, b in bucket_slices(hi - lo, bucket_elems, lo):
        w = dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def average_gradients(flat_grad: torch.Tensor, n_trainable: int, group=None, bucket_elems: int = 16 << 20, async_op: bool = False):
    """In-place MEAN over ranks of flat_grad[:n_trainable] (for callers that run a stock optimizer; the fused trainers use
    sum_gradients + the optimizer kernel's grad_scale instead, which saves the scaling pass)."""
    world = _world(group)
    if world <= 1:
        return []
    flat_grad[:n_trainable].mul_(1.0 / world)
    return sum_gradients(flat_grad, 0, n_trainable, group, bucket_elems, async_op)


class GradientExchange:
    """The exchange schedule of one optimizer step:  start_early() at the backward's cut mark, finish() after the backward.
    finish() returns once every all-reduce has been waited for (for NCCL/RCCL `wait` only makes the current stream wait: the host
    does not block).  With world 1 both calls are no-ops."""

    def __init__(self, flat_grad: torch.Tensor, n_late: int, n_trainable: int, group=None, bucket_elems: int = 16 << 20, force: bool = False):
        if not 0 <= n_late <= n_trainable <= flat_grad.numel():
            raise ValueError("need 0 <= n_late <= n_trainable <= numel")
        self.grad, self.n_late, self.n_trainable, self.group, self.bucket = flat_grad, n_late, n_trainable, group, bucket_elems
        self.force = force
        self.works = []
        self.early_started = False

    def start_early(self):
        assert not self.early_started
        self.early_started = True
        self.works += sum_gradients(self.grad, self.n_late, self.n_trainable, self.group, self.bucket, async_op=True, force=self.force)

    def finish(self):
        lo_hi = (0, self.n_late) if self.early_started else (0, self.n_trainable)
        self.works += sum_gradients(self.grad, lo_hi[0], lo_hi[1], self.group, self.bucket, async_op=True, force=self.force)
        for w in self.works:
            w.wait()
        self.works, self.early_started = [], False


def broadcast_parameters(flat_data: torch.Tensor, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters (whole arena, including the unused tail)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)
