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


def test_sync_bn_two_half_batches_are_the_whole_batch():
    """VERDICT r02 item 5.  The reference normalises over its single batch of 32 (CAE_improved_modeltrain.py:192-213, batch_size=32
    at :287).  Two trainers in ONE process, each given half of a batch, exchange their BatchNormalization partials through a
    fake communicator (cs_train_set_sync_bn: the all-gather is two threads swapping slots of their exchange buffers): the mean
    of their two gradients, their losses and their moving statistics must be those of the single batch-32 step."""
    import threading
    import torch
    from cellscreen.trainer import Trainer, param_layout, split_flat
    w = synth.random_cae(seed=11)
    y = np.concatenate([synth.blob_crops(1, 16), synth.synth_crops(1, 0, 16)])
    x = np.clip(y + 0.02 * np.random.default_rng(1).standard_normal(y.shape).astype(np.float32), 0, 1).astype(np.float32)
    # the two halves differ in kind (blobs | noise): per-rank statistics would be far from the batch's
    one = Trainer(w)
    loss1, mae1 = one.forward_backward(x, y)
    _, mov1, g1 = one.export_flat(grads=True)
    one.close()
    tr = [Trainer(w), Trainer(w)]
    barrier = threading.Barrier(2)
    bufs = {}

    def communicator(rank):
        def all_gather(buf, fpr):
            bufs[rank] = buf
            barrier.wait()                                   # both slots are written (each library drained its stream first)
            o = 1 - rank
            buf[o * fpr:(o + 1) * fpr].copy_(bufs[o][o * fpr:(o + 1) * fpr])
            torch.cuda.synchronize()
            barrier.wait()                                   # nobody rewrites its slot before the other has copied it
        return all_gather
    out = [None, None]

    def run(rank):
        tr[rank].set_sync_bn(communicator(rank), rank, 2)
        out[rank] = tr[rank].forward_backward(x[16 * rank:16 * rank + 16], y[16 * rank:16 * rank + 16])
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    try:
        assert out[0] is not None and out[1] is not None
        ex = [t.export_flat(grads=True) for t in tr]
        assert abs((out[0][0] + out[1][0]) / 2 - loss1) <= 1e-5 * loss1 and abs((out[0][1] + out[1][1]) / 2 - mae1) <= 1e-5 * mae1
        assert np.array_equal(ex[0][1], ex[1][1])                          # ONE set of moving statistics ...
        assert np.allclose(ex[0][1], mov1, rtol=1e-5, atol=1e-7)           # ... the single process's
        ga = split_flat((ex[0][2].astype(np.float64) + ex[1][2]) / 2, param_layout())
        gr = split_flat(g1.astype(np.float64), param_layout())
        errs = {k: np.linalg.norm(ga[k] - gr[k]) / max(np.linalg.norm(gr[k]), 1e-30) for k in gr}
        print("sync-BN: mean of two half-batch gradients vs the batch-32 gradient, relative L2:", {k: float("%.1e" % v) for k, v in errs.items()})
        assert max(errs.values()) <= 1e-5, errs
        # without the exchange the halves normalise by themselves: a different gradient (the deviation sync_bn removes)
        solo = [Trainer(w), Trainer(w)]
        for r in range(2):
            solo[r].forward_backward(x[16 * r:16 * r + 16], y[16 * r:16 * r + 16])
        gs = split_flat((solo[0].export_flat(grads=True)[2].astype(np.float64) + solo[1].export_flat(grads=True)[2]) / 2, param_layout())
        for t in solo:
            t.close()
        assert max(np.linalg.norm(gs[k] - gr[k]) / max(np.linalg.norm(gr[k]), 1e-30) for k in gr) > 1e-3
    finally:
        for t in tr:
            t.close()
