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


# ---- data-parallel training loop: the ranks must take every callback decision together ------------------------------
class _RankTrainer:
    """Stands in for the GPU trainer in cellscreen.training (as tests/test_callbacks_cpu.py does), with a validation loss that
    DIFFERS per rank -- what per-rank BatchNormalization moving statistics produce."""
    script = None
    rank = 0
    instances = []

    def __init__(self, init, device_id=0):
        self.n_trainable, self.n_moving = 4, 2
        self.epoch, self.lrs, self.moving = 0, [], np.full(2, float(_RankTrainer.rank), np.float32)
        self.moving_seen_at_eval = []
        _RankTrainer.instances.append(self)

    def use_grad_tensor(self, t):
        self.g = t

    def enable_sync_bn(self, dist, rank, world):
        self.sync = (rank, world)

    def forward_backward(self, x, y):
        self.g.fill_(1.0 + _RankTrainer.rank)
        return 0.1 * (1 + _RankTrainer.rank), 0.2

    def apply(self, lr):
        self.lrs.append(lr)
        assert float(self.g[0]) == 1.5                 # the mean of the two ranks' gradients

    def evaluate(self, x, y):
        self.moving_seen_at_eval.append(self.moving.copy())
        v = _RankTrainer.script[_RankTrainer.rank][self.epoch]
        self.epoch += 1
        return v, v

    def export_flat(self):
        return np.full(4, self.epoch - 1, np.float32), self.moving.copy()

    def load_flat(self, p, m):
        if m is not None:
            self.moving = np.asarray(m, np.float32).copy()

    def weights(self):
        from cellscreen import synth
        return synth.random_cae(seed=1, trivial_bn=True)

    def close(self):
        pass


def _train_worker(rank, world, port, outdir, q):
    import torch
    import torch.distributed as dist
    from cellscreen import training
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    training.Trainer = _RankTrainer
    _RankTrainer.rank = rank
    # rank 0 improves for 4 epochs and then stalls; rank 1's own numbers would keep improving for ever
    _RankTrainer.script = {0: [1.0, 0.8, 0.6, 0.5] + [0.55] * 40, 1: [1.0 - 0.01 * e for e in range(44)]}
    t = training.ImprovedAnomalyDetectionTraining(os.path.join(outdir, f"r{rank}"), epochs=40, verbose=0, augment=None, data_parallel=True)
    _, _, hist = t.train_autoencoder(np.zeros((100, 64, 64), np.float32))
    tr = _RankTrainer.instances[-1]
    q.put((rank, hist.stopped_epoch, list(hist.lr_reduced_epochs), hist.history["val_loss"], tr.lrs, [m.tolist() for m in tr.moving_seen_at_eval],
           getattr(tr, "sync", None)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ranks_stop_and_halve_the_rate_together(tmp_path):
    """World 2 on gloo: scripted val_loss values that diverge between the ranks.  Every rank must act on rank 0's number -- the
    same stop epoch, the same epochs of learning-rate halving, the same rate at every step -- and validate with ONE set of
    moving statistics (the mean over the ranks)."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (_, stop0, red0, val0, lrs0, mov0, sync0), (_, stop1, red1, val1, lrs1, mov1, sync1) = got
    assert stop0 == stop1 == 13 and red0 == red1 == [8, 13]          # rank 0's script (tests/test_callbacks_cpu.py has the same one)
    assert val0 == val1 and lrs0 == lrs1 and len(lrs0) == 14 * (80 // 32)
    assert mov0 == mov1 and mov0[0] == [0.5, 0.5]                       # the mean of the ranks' statistics (0 and 1)
    assert sync0 == (0, 2) and sync1 == (1, 2)                          # synchronised BatchNormalization is the default
