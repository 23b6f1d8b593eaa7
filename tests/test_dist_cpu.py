"""world_size-2 test of the sharded gather on the gloo backend (CPU): the N>1 path of
bench.py / cellscreen.dist without a GPU."""
import os
import socket

import numpy as np
import pytest

import helpers as H  # noqa: F401
from cellscreen import dist as csdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = csdist.shard_range(n_total, rank, world)
    idx = np.arange(lo, hi)
    local = dict(mse=(idx * 0.5).astype(np.float32), mae=(idx * 0.25).astype(np.float32),
                 cons_score=idx.astype(np.float64) - 3.0, mod_score=-idx.astype(np.float64),
                 cons_pred=np.where(idx % 3 == 0, -1, 1).astype(np.int8), mod_pred=np.where(idx % 2 == 0, -1, 1).astype(np.int8))
    out = csdist.gather_results(csdist.to_torch(local), n_total, dst=0)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 10])
def test_gather_world2_gloo(n_total):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    idx = np.arange(n_total)
    assert np.array_equal(got["mse"], (idx * 0.5).astype(np.float32))
    assert np.array_equal(got["cons_score"], idx.astype(np.float64) - 3.0)
    assert np.array_equal(got["cons_pred"], np.where(idx % 3 == 0, -1, 1).astype(np.int8))
    assert got["mod_pred"].dtype == np.int8 and len(got["mod_score"]) == n_total


def _grad_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(84289, dtype=torch.float32) * (rank + 1)
    csdist.allreduce_mean_(g)
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_mean_world2_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got, np.arange(84289, dtype=np.float32) * 1.5)
