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


def _expected(n_total):
    idx = np.arange(n_total)
    return dict(mse=(idx * 0.5).astype(np.float32), mae=(idx * 0.25).astype(np.float32),
                cons_score=idx.astype(np.float64) - 3.0, mod_score=-idx.astype(np.float64),
                cons_pred=np.where(idx % 3 == 0, -1, 1).astype(np.int8), mod_pred=np.where(idx % 2 == 0, -1, 1).astype(np.int8))


def test_record_packing_round_trips_every_field():
    torch = pytest.importorskip("torch")
    want = _expected(1001)
    want["cons_score"][5] = -0.0
    want["mod_score"][6] = 1e-300
    rec = csdist.pack_records(csdist.to_torch(want), rows=1024)
    assert rec.shape == (1024, csdist.RECORD_BYTES) and rec.dtype == torch.uint8 and int(rec[1001:].sum()) == 0
    back = csdist.unpack_records(rec, 1001)
    for k, v in want.items():
        assert back[k].numpy().dtype == v.dtype and np.array_equal(back[k].numpy().view(np.uint8), v.view(np.uint8)), k


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
    every = csdist.gather_results(csdist.to_torch(local), n_total)          # dst=None: all ranks get the global arrays
    assert all(torch.equal(every[k], torch.as_tensor(v)) for k, v in _expected(n_total).items()), "all-gather form"
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
    for k, v in _expected(n_total).items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k


def _real_gradient(rank):
    """A REAL gradient of the reference graph: one training-mode forward + backward of the numpy train oracle on this
    rank's own batch, flattened in the trainer's parameter order (cs_train_export's layout, trainer.param_layout)."""
    from cellscreen import synth
    from cellscreen.trainer import param_layout
    from oracle import train_oracle as T
    st = T.TrainState(synth.random_cae(seed=42, trivial_bn=True), dtype=np.float32)
    x = synth.blob_crops(100 + rank, 4)
    r = T.forward_backward(st, x, x)
    flat = np.concatenate([np.asarray(g, np.float32).ravel() for g in r["grads"]])
    assert flat.size == sum(int(np.prod(s)) for _, s in param_layout()) == 84289
    return flat


def _grad_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.from_numpy(_real_gradient(rank).copy())
    csdist.allreduce_mean_(g)
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_mean_world2_gloo():
    """The data-parallel training exchange (cs_train_forward_backward -> all-reduce -> cs_train_apply) on two real
    gradients: every rank ends with their element-wise mean."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    g0, g1 = _real_gradient(0), _real_gradient(1)
    assert np.abs(g0).max() > 0 and not np.array_equal(g0, g1)
    assert np.array_equal(got, (g0 + g1) / np.float32(2))
