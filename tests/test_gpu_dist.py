"""World-size-1 RCCL smoke on the GPU box: the N > 1 code paths -- the packed one-collective
gather of per-cell results and the data-parallel training exchange (forward_backward -> all-reduce -> apply on a
gradient buffer that is a torch tensor) -- run through backend "nccl" (= RCCL) with a real gradient and must equal the
single-process calls bit for bit.  The multi-rank arithmetic is covered by the gloo tests (tests/test_dist_cpu.py)."""
import os
import socket

import numpy as np
import pytest

import helpers as H  # noqa: F401
from cellscreen import dist as csdist
from cellscreen import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_packed_gather_over_rccl(nccl_world1, golden_det):
    import torch
    from cellscreen.engine import Engine
    det = H.det_from_golden(golden_det)
    e = Engine.from_weights(synth.random_cae(seed=42), None, det)
    x = torch.from_numpy(synth.synth_crops(3, 0, 1000)).cuda()
    r = e.screen(x)
    for dst in (0, None):
        g = csdist.gather_results(r, 1000, dst=dst)
        for k in r:
            assert g[k].dtype == r[k].dtype and torch.equal(g[k], r[k]), (dst, k)
    e.close()


def test_data_parallel_step_equals_the_fused_step(nccl_world1):
    """Trainer.use_grad_tensor + forward_backward + allreduce_mean_ (async on RCCL's stream; torch orders its current
    stream after it) + apply -- with NO host synchronisation in between -- against cs_train_step on a twin trainer."""
    import torch
    from cellscreen.trainer import Trainer
    w = synth.random_cae(seed=42, trivial_bn=True)
    a, b = Trainer(w), Trainer(w)
    g = torch.zeros(a.n_trainable, dtype=torch.float32, device="cuda")
    a.use_grad_tensor(g)
    X = torch.from_numpy(synth.blob_crops(7, 256)).cuda()
    for s in range(6):
        idx = torch.randint(0, 256, (32,), device="cuda")
        xb = X[idx]                                          # produced asynchronously on torch's stream
        la, ma = a.forward_backward(xb, xb)
        csdist.allreduce_mean_(g)
        a.apply(1e-3)
        lb, mb = b.step(xb, xb, 1e-3)
        assert la == lb and ma == mb, s
    pa, ma_ = a.export_flat()
    pb, mb_ = b.export_flat()
    assert np.array_equal(pa, pb) and np.array_equal(ma_, mb_)
    _, _, ga = a.export_flat(grads=True)
    assert np.array_equal(ga, g.cpu().numpy()) and np.abs(ga).max() > 0
    a.close(); b.close()


def test_training_class_data_parallel_world1(nccl_world1, tmp_path):
    """ImprovedAnomalyDetectionTraining(data_parallel=True) at world size 1 is the single-process run."""
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    cells = synth.blob_crops(23, 320)
    t1 = ImprovedAnomalyDetectionTraining(str(tmp_path / "dp"), epochs=2, verbose=0, augment=None, data_parallel=True)
    t2 = ImprovedAnomalyDetectionTraining(str(tmp_path / "sp"), epochs=2, verbose=0, augment=None)
    a1, _, h1 = t1.train_autoencoder(cells)
    a2, _, h2 = t2.train_autoencoder(cells)
    assert h1.history["loss"] == h2.history["loss"] and h1.history["val_loss"] == h2.history["val_loss"]
    assert all(np.array_equal(x, y) for x, y in zip(a1.kernels, a2.kernels))
