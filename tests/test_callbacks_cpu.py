"""The callback state machine (cellscreen/callbacks.py) against the rules of the three Keras callbacks the reference
passes to fit() (CAE_improved_modeltrain.py:263-283), driven by scripted val_loss sequences -- and the training loop of
cellscreen/training.py driven through a stub trainer, so that patience, LR halving, the stop epoch and which weights
end up in the returned model are all asserted without a GPU."""
import numpy as np
import pytest

from cellscreen import spec
from cellscreen.callbacks import FitCallbacks


def run(vals, **kw):
    cb = FitCallbacks(**kw)
    acts = []
    for e, v in enumerate(vals):
        a = cb.on_epoch_end(e, v)
        acts.append(a)
        if a.stop_training:
            break
    return cb, acts


def test_early_stopping_patience_and_best_epoch():
    # improves for 3 epochs, then flat: stop at the end of the 10th non-improving epoch (epoch index 12)
    vals = [1.0, 0.9, 0.8] + [0.85] * 30
    cb, acts = run(vals)
    assert len(acts) == 13 and acts[-1].stop_training and cb.stopped_epoch == 12
    assert cb.es_best_epoch == 2 and cb.es_best == 0.8
    assert [a.snapshot_best_weights for a in acts[:4]] == [True, True, True, False]
    assert cb.restore_best_at_train_end()


def test_early_stopping_min_delta_is_zero_any_decrease_counts():
    vals = [1.0 - 1e-9 * i for i in range(40)]              # strictly decreasing by 1e-9: never stops
    cb, acts = run(vals)
    assert len(acts) == 40 and not any(a.stop_training for a in acts) and cb.es_best_epoch == 39


def test_reduce_lr_on_plateau_needs_1e_4_improvement():
    """Keras's default min_delta = 1e-4: with val_loss creeping down by 1e-6 per epoch EarlyStopping sees improvement
    every epoch, ReduceLROnPlateau sees none after the first and halves the rate every 5 epochs."""
    vals = [1e-3 - 1e-6 * i for i in range(21)]
    cb, acts = run(vals)
    assert not any(a.stop_training for a in acts)
    assert cb.lr_reduced_epochs == [5, 10, 15, 20]          # epoch 0 sets best; waits at epochs 1..5 -> reduce at 5; ...
    lr0 = float(np.float32(1e-3))
    assert [a.lr for a in acts][4:7] == [lr0, float(np.float32(lr0 * 0.5)), float(np.float32(lr0 * 0.5))]
    assert cb.rl_best == vals[0]                            # best is only replaced by a >= 1e-4 improvement
    # ... and once the accumulated decrease passes 1e-4 it IS an improvement: -1e-5 per epoch -> reset at epoch 11
    cb2, _ = run([1e-3 - 1e-5 * i for i in range(21)])
    assert cb2.lr_reduced_epochs == [5, 10, 16] and cb2.rl_best == pytest.approx(1e-3 - 11e-5)


def test_reduce_lr_improvement_resets_wait():
    vals = [1.0, 1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    cb, acts = run(vals)
    assert cb.lr_reduced_epochs == [9]                      # waits 1..3, reset at epoch 4, waits 5..9 -> reduce at 9


def test_reduce_lr_stops_at_min_lr_and_wait_is_not_reset_there():
    cb = FitCallbacks(es_patience=10 ** 6)
    cb.on_epoch_end(0, 1.0)
    lrs = []
    for e in range(1, 80):
        lrs.append(cb.on_epoch_end(e, 1.0).lr)
    assert min(lrs) == pytest.approx(float(np.float32(spec.RLROP_MIN_LR)))
    floor_epoch = cb.lr_reduced_epochs[-1]
    assert len(cb.lr_reduced_epochs) == 10                  # 1e-3 * 0.5^10 < 1e-6: the 10th reduction clamps to min_lr
    assert cb.lr_reduced_epochs == list(range(5, 55, 5))
    assert cb.rl_wait == 79 - floor_epoch                   # no reduction, no reset once lr == min_lr


def test_model_checkpoint_saves_only_strict_improvements_and_never_nan():
    cb, acts = run([0.5, 0.5, 0.4, float("nan"), 0.45, 0.3])
    assert [a.save_checkpoint for a in acts] == [True, False, True, False, False, True]
    assert cb.mc_saved_epochs == [0, 2, 5]


def test_nan_first_epoch_is_taken_as_best_by_early_stopping_like_keras():
    cb, acts = run([float("nan")] + [1.0] * 12)
    # best = nan after epoch 0 (best was None); np.less(x, nan) is False for ever after: stops at epoch 10
    assert acts[0].snapshot_best_weights and cb.stopped_epoch == 10 and cb.es_best_epoch == 0
    assert not any(a.save_checkpoint for a in acts[:1])


def test_keras2_restores_only_when_the_stop_fired():
    vals = [1.0, 0.5, 0.6, 0.7]
    cb3, _ = run(vals, keras_version=3)
    cb2, _ = run(vals, keras_version=2)
    assert cb3.restore_best_at_train_end() and not cb2.restore_best_at_train_end()
    cb2b, acts = run([1.0] + [2.0] * 20, keras_version=2)
    assert acts[-1].stop_training and cb2b.restore_best_at_train_end()


# ---- the training loop itself, with a stub in place of the GPU trainer ---------------------------------------------
class _StubTrainer:
    """Counts steps; `weights` is a single number = the epoch it was exported in; val_loss follows a script."""
    script = []
    instances = []

    def __init__(self, init, device_id=0):
        self.n_trainable, self.n_moving = 4, 2
        self.epoch_steps, self.epoch, self.loaded = 0, 0, None
        self.lrs = []
        _StubTrainer.instances.append(self)

    def step(self, x, y, lr):
        self.lrs.append(lr)
        return 0.1, 0.1

    def step_async(self, x, y, lr):
        self.lrs.append(lr)
        self.nsteps = getattr(self, "nsteps", 0) + 1

    def read_metrics(self, reset=True):
        n, self.nsteps = getattr(self, "nsteps", 0), 0
        return 0.1, 0.1, n

    def evaluate(self, x, y):
        v = _StubTrainer.script[self.epoch]
        self.epoch += 1
        return v, v

    def export_flat(self):
        return np.full(4, self.epoch - 1, np.float32), np.zeros(2, np.float32)

    def load_flat(self, p, m):
        self.loaded = int(p[0])

    def weights(self):
        from cellscreen import synth
        w = synth.random_cae(seed=1, trivial_bn=True)
        w.biases[0][:] = self.loaded if self.loaded is not None else self.epoch - 1     # tag: which epoch's weights these are
        return w

    def close(self):
        pass


@pytest.fixture
def stub_training(monkeypatch, tmp_path):
    import types
    import sys
    from cellscreen import training
    monkeypatch.setattr(training, "Trainer", _StubTrainer)
    fake_torch = types.SimpleNamespace(
        cuda=types.SimpleNamespace(is_available=lambda: True),
        device=lambda *a: "dev",
        from_numpy=lambda a: _FakeTensor(a))
    monkeypatch.setitem(sys.modules, "torch", fake_torch)
    _StubTrainer.instances.clear()
    return training.ImprovedAnomalyDetectionTraining(str(tmp_path / "out"), epochs=40, verbose=0, augment=None)


class _FakeTensor:
    def __init__(self, a):
        self.a = np.asarray(a)

    def to(self, dev):
        return self

    def __getitem__(self, i):
        return _FakeTensor(self.a[i.a if isinstance(i, _FakeTensor) else i])

    def __len__(self):
        return len(self.a)


def test_training_loop_stop_epoch_lr_schedule_and_restored_weights(stub_training):
    from cellscreen import model_io
    import os
    # best at epoch 3 (0-based), then worse for ever: LR halves at epochs 8 and 13, stop at 13
    _StubTrainer.script = [1.0, 0.8, 0.6, 0.5] + [0.55] * 40
    cells = np.zeros((100, 64, 64), np.float32)
    ae, enc, hist = stub_training.train_autoencoder(cells)
    tr = _StubTrainer.instances[-1]
    assert hist.stopped_epoch == 13 and len(hist.history["val_loss"]) == 14
    assert hist.lr_reduced_epochs == [8, 13]
    lr0 = float(np.float32(1e-3))
    steps = 80 // 32
    assert tr.lrs[:9 * steps] == [lr0] * (9 * steps) and tr.lrs[9 * steps] == lr0 / 2
    assert hist.best_epoch == 3 and tr.loaded == 3                                 # restore_best_weights=True
    assert ae.biases[0][0] == 3.0 and enc.n_conv == 3
    out = stub_training.output_dir
    best = model_io.cae_from_keras(os.path.join(out, "best_autoencoder.keras"))    # ModelCheckpoint's last save: epoch 3
    final = model_io.cae_from_keras(os.path.join(out, "final_autoencoder.keras"))
    enck = model_io.cae_from_keras(os.path.join(out, "encoder.keras"))
    assert best.biases[0][0] == 3.0 and final.biases[0][0] == 3.0 and enck.n_conv == 3


def test_training_loop_keras2_keeps_last_epoch_weights_without_a_stop(stub_training):
    _StubTrainer.script = [1.0, 0.5, 0.6, 0.7, 0.8] + [0.9] * 40
    stub_training.epochs = 5
    stub_training.keras_version = 2
    ae, _, hist = stub_training.train_autoencoder(np.zeros((100, 64, 64), np.float32))
    assert hist.stopped_epoch is None and ae.biases[0][0] == 4.0                    # last epoch's weights
    stub_training.keras_version = 3
    ae3, _, _ = stub_training.train_autoencoder(np.zeros((100, 64, 64), np.float32))
    assert ae3.biases[0][0] == 1.0                                                  # Keras 3: best epoch restored at train end
