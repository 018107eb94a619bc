"""N > 1 path on CPU: world_size-2 gloo run of the flat-arena gradient averaging and parameter broadcast that
DDPMTrainer uses between its forward/backward graph and its optimizer graph."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from medical_image_generation_amd import ddp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, n_trainable, bucket, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        works = ddp.average_gradients(g, n_trainable, bucket_elems=bucket, async_op=(rank == 0 or True))
        for w in works:
            w.wait()
        data = torch.full((n,), float(rank + 5))
        ddp.broadcast_parameters(data, 0)
        out[rank] = (g.clone(), data.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket", [7, 64, 1 << 20])
def test_average_gradients_two_ranks(bucket):
    world, n, nt = 2, 1000, 900
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, nt, bucket, out), nprocs=world, join=True)
    base = torch.arange(n, dtype=torch.float32)
    for rank in range(world):
        g, data = out[rank]
        assert torch.allclose(g[:nt], base[:nt] * 1.5)           # mean of 1x and 2x
        assert torch.equal(g[nt:], base[nt:] * (rank + 1))      # unused tail never communicated
        assert torch.equal(data, torch.full((n,), 5.0))          # rank 0's parameters everywhere


def _worker_exchange(rank, world, port, n, n_late, nt, bucket, out):
    """The trainer's reduce-then-step sequence on a fake arena: 'backward' writes the early segment, the exchange starts at the cut,
    the rest of the 'backward' writes the late prefix while the early all-reduce is in flight, finish(), then an Adam step with the
    1/world factor folded in as grad_scale (what mi_adam_step does on the GPU)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(100 + rank)  # every rank has its own micro-batch -> its own gradient
        full = torch.randn(n, generator=gen)
        params = torch.full((n,), float(rank))            # ranks start apart ...
        ddp.broadcast_parameters(params, 0)               # ... and begin from rank 0's parameters
        grad = torch.full((n,), float("nan"))
        grad[nt:] = 7.0                                    # statically unused tail (proj_attn.*): never written, never communicated
        ex = ddp.GradientExchange(grad, n_late, nt, bucket_elems=bucket)
        grad[n_late:nt] = full[n_late:nt]                  # backward up to the cut: early segment final
        ex.start_early()
        grad[:n_late] = full[:n_late]                      # rest of the backward, overlapping the exchange
        ex.finish()
        assert ex.works == [] and not ex.early_started
        g = grad[:nt] * (1.0 / world)                      # grad_scale
        clip = min(1.0, 1.0 / (float(g.norm()) + 1e-6))
        m = 0.1 * g * clip
        v = 0.001 * (g * clip) ** 2
        params[:nt] -= 1e-3 * (m / 0.1) / ((v / 0.001).sqrt() + 1e-8)
        out[rank] = (grad.clone(), params.clone(), full.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket,n_late", [(64, 100), (1 << 20, 100), (37, 0), (50, 900)])
def test_gradient_exchange_two_ranks(bucket, n_late):
    """world_size-2 gloo run of the overlapped exchange: both ranks end with bit-identical gradients (= the SUM of the two ranks'
    gradients on the trainable prefix), bit-identical parameters after the step, and an untouched unused tail."""
    world, n, nt = 2, 1000, 900
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_exchange, args=(world, port, n, n_late, nt, bucket, out), nprocs=world, join=True)
    (g0, p0, f0), (g1, p1, f1) = out[0], out[1]
    assert torch.equal(g0[:nt], g1[:nt]) and torch.allclose(g0[:nt], (f0 + f1)[:nt])
    assert torch.equal(g0[nt:], torch.full((n - nt,), 7.0)) and torch.equal(g1[nt:], g0[nt:])
    assert torch.equal(p0, p1) and not torch.equal(p0[:nt], torch.zeros(nt))
    assert torch.equal(p0[nt:], torch.zeros(n - nt))  # rank 0's (zero) parameters, never stepped


def test_gradient_exchange_validates_and_single_process_is_noop():
    g = torch.ones(10)
    ex = ddp.GradientExchange(g, 3, 8)
    ex.start_early()
    ex.finish()
    assert torch.equal(g, torch.ones(10))
    with pytest.raises(ValueError):
        ddp.GradientExchange(g, 9, 8)
    assert ddp.sum_gradients(g, 0, 8) == []


def test_arena_orders_late_gradients_first():
    """The C4 net's arena: [late | early | untrainable]; the late prefix holds exactly the tensors whose gradients complete after the
    backward's cut mark (finest down levels, conv_in, time-embedding MLP, every time_emb_proj and conv1 bias) and is a few percent."""
    import bench
    from medical_image_generation_amd import engine as E
    from medical_image_generation_amd.unet import DiffusionModelUNet
    net = DiffusionModelUNet(**bench.C4)
    a = E.ParamArena(net._entries, "cpu", late=net._late)
    assert 0 < a.n_late < 0.05 * a.n_trainable and a.n_trainable < a.numel
    for name, off in a.offsets.items():
        is_late = (name.startswith(("down_blocks.0.", "down_blocks.1.", "conv_in.", "time_embed.")) or ".time_emb_proj." in name
                   or name.endswith(".conv1.conv.bias")) and ".proj_attn." not in name
        assert (off < a.n_late) == is_late, name
    # adjacency the kernels rely on survived the reordering
    a.span([r[0] + ".time_emb_proj.weight" for r in net._resnets])
    a.span([f"middle_block.attention.to_{t}.weight" for t in "qkv"])


def test_bucket_slices_cover_exactly():
    for n, b in [(10, 3), (9, 3), (1, 100), (0, 5)]:
        sl = ddp.bucket_slices(n, b)
        assert [a for a, _ in sl] == list(range(0, n, b)) and (not sl or sl[-1][1] == n)
        assert all(e - a <= b for a, e in sl)
    assert ddp.bucket_slices(10, 4, start=5) == [(5, 9), (9, 13), (13, 15)]
    with pytest.raises(ValueError):
        ddp.bucket_slices(10, 0)


def test_single_process_is_a_no_op():
    g = torch.ones(8)
    assert ddp.average_gradients(g, 8) == [] and torch.equal(g, torch.ones(8))
