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


def test_bucket_slices_cover_exactly():
    for n, b in [(10, 3), (9, 3), (1, 100), (0, 5)]:
        sl = ddp.bucket_slices(n, b)
        assert [a for a, _ in sl] == list(range(0, n, b)) and (not sl or sl[-1][1] == n)
        assert all(e - a <= b for a, e in sl)
    with pytest.raises(ValueError):
        ddp.bucket_slices(10, 0)


def test_single_process_is_a_no_op():
    g = torch.ones(8)
    assert ddp.average_gradients(g, 8) == [] and torch.equal(g, torch.ones(8))
