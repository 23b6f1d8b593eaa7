"""Training-step parity: the HIP trainer through the C ABI
against the numpy oracle (oracle/train_oracle.py, pinned to torch autograd) on the same batch.
Stated tolerance (Appendix G): per-tensor gradient relative L2 <= 1e-5; loss relative 1e-5."""
import numpy as np
import pytest

import helpers as H  # noqa: F401
from cellscreen import spec, synth
from cellscreen.trainer import Trainer, flat_from_weights, param_layout, split_flat
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu
TOL_GRAD = 1e-5


def batch(n, seed=1):
    y = np.concatenate([synth.blob_crops(seed, n // 2), synth.synth_crops(seed, 0, n - n // 2)])
    rng = np.random.default_rng(seed)
    x = np.clip(y + 0.02 * rng.standard_normal(y.shape).astype(np.float32), 0, 1).astype(np.float32)   # augmented input != target
    return x, y


def grads_by_name(flat):
    return split_flat(flat, param_layout())


def activation_pattern(tr, w, n):
    """The trainer's ReLU masks and max-pool routing, from its relu outputs (stage tap 0).
    BN is monotone in r (increasing for gamma > 0, decreasing for gamma < 0), so the arg-max of
    BN(r) over a window is the first arg-max of sign(gamma) * r."""
    masks, args = [], []
    for l in range(6):
        r = tr.tensor(0, l, n)
        masks.append(r > 0)
        if l < 3:
            N, Hh, Ww, C = r.shape
            win = r.reshape(N, Hh // 2, 2, Ww // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(N, Hh // 2, Ww // 2, C, 4)
            args.append(np.argmax(win * np.sign(w.bn_gamma[l])[None, None, None, :, None], axis=-1))
        else:
            args.append(None)
    return masks + [None], args + [None] * 4


@pytest.mark.parametrize("n", [32, 5])
def test_forward_backward_gradients(n):
    w = synth.random_cae(seed=11)
    x, y = batch(n)
    tr = Trainer(w)
    loss, mae = tr.forward_backward(x, y)
    # ReLU' and max-pool routing are discontinuous: an fp32 and an fp64 evaluation disagree on a
    # handful of the ~1e6 decisions per layer, so the oracle is run on the trainer's pattern
    masks, args = activation_pattern(tr, w, n)
    st = T.TrainState(w, dtype=np.float64)
    ref = T.forward_backward(st, x, y, relu_masks=masks, pool_args=args)
    free = T.forward_backward(T.TrainState(w, dtype=np.float64), x, y)
    flips = sum(int(np.sum(m != (r > 0))) for m, r in zip(masks[:6], free["relu"][:6]))
    print("ReLU decisions that differ from the unconstrained fp64 oracle:", flips, "of", sum(m.size for m in masks[:6]))
    assert flips <= 1e-5 * sum(m.size for m in masks[:6])
    assert abs(loss - ref["loss"]) <= 1e-5 * ref["loss"], (loss, ref["loss"])
    assert abs(mae - ref["mae"]) <= 1e-5 * ref["mae"]
    _, mov, g = tr.export_flat(grads=True)
    got = grads_by_name(g)
    errs = {}
    for (name, _shape), gr in zip(param_layout(), ref["grads"]):
        errs[name] = np.linalg.norm(got[name].astype(np.float64) - gr) / max(np.linalg.norm(gr), 1e-30)
    print("gradient relative L2 errors:", {k: float("%.2e" % v) for k, v in errs.items()})
    for name, err in errs.items():
        assert err <= TOL_GRAD, f"{name}: relative L2 error {err:.3e}" 
    # moving statistics: momentum 0.99 update with the batch mean / biased variance
    o = 0
    for l in range(6):
        c = spec.CHANNELS[l]
        assert np.allclose(mov[o:o + c], st.mov_mean[l], rtol=1e-5, atol=1e-7); o += c
        assert np.allclose(mov[o:o + c], st.mov_var[l], rtol=1e-5, atol=1e-7); o += c
    tr.close()


def test_gradients_against_the_unconstrained_oracle_on_a_batch_with_margins():
    """one gradient check in which the oracle runs FREE -- it takes its own ReLU and max-pool decisions, nothing
    is handed over from the trainer.  That is only meaningful on a batch whose every decision is safe from rounding: weights
    seed 32 / blob crop 51 (found by a seeded search, oracle/train_oracle.py reports the margins) has every pre-activation at least
    1.8e-6 of its layer's range away from 0 and every live pooling window's winner at least 3.9e-6 of the range ahead of its
    runner-up -- the fp32 pre-activations are good to ~1e-7 (conv1, K = 9) .. 5e-7 (K = 288 / 576) of the range.  The margins are
    asserted, then all 26 gradient tensors, the loss and the MAE must agree at the usual bars with NO pattern shared."""
    w = synth.random_cae(seed=32)
    y = synth.blob_crops(51, 1)
    x = np.clip(y + 0.02 * np.random.default_rng(51).standard_normal(y.shape).astype(np.float32), 0, 1).astype(np.float32)
    ref = T.forward_backward(T.TrainState(w, dtype=np.float64), x, y, update_moving=False)
    worst = min(min(m.values()) for m in ref["margins"])
    assert worst >= 1.5e-6, ref["margins"]
    tr = Trainer(w)
    try:
        loss, mae = tr.forward_backward(x, y)
        masks, _ = activation_pattern(tr, w, 1)
        assert all(np.array_equal(m, r > 0) for m, r in zip(masks[:6], ref["relu"][:6]))      # the same decisions, taken independently
        assert abs(loss - ref["loss"]) <= 1e-5 * ref["loss"] and abs(mae - ref["mae"]) <= 1e-5 * ref["mae"]
        _, _, g = tr.export_flat(grads=True)
        got = grads_by_name(g)
        errs = {name: np.linalg.norm(got[name].astype(np.float64) - gr) / max(np.linalg.norm(gr), 1e-30)
                for (name, _shape), gr in zip(param_layout(), ref["grads"])}
        print("free-oracle gradient relative L2 errors:", {k: float("%.2e" % v) for k, v in errs.items()})
        for name, err in errs.items():
            assert err <= TOL_GRAD, f"{name}: relative L2 error {err:.3e}"
    finally:
        tr.close()


def test_adam_update_is_keras_formula():
    """cs_train_apply on the trainer's own gradients == Keras Adam evaluated in float64:
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m, v EMAs; w -= alpha*m/(sqrt(v)+1e-7)."""
    w = synth.random_cae(seed=12, trivial_bn=True)      # Keras init: gamma 1, beta 0, mean 0, var 1
    x, y = batch(16, seed=3)
    tr = Trainer(w)
    p, _ = flat_from_weights(w)
    p = p.astype(np.float64)
    m = np.zeros_like(p); v = np.zeros_like(p)
    for t in range(1, 4):
        tr.forward_backward(x, y)
        _, _, g = tr.export_flat(grads=True)
        tr.apply(lr=1e-3)
        g = g.astype(np.float64)
        m += (g - m) * (1 - 0.9); v += (g * g - v) * (1 - 0.999)
        alpha = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        p = p - alpha * m / (np.sqrt(v) + 1e-7)
        got, _ = tr.export_flat()
        assert np.abs(got - p).max() <= 2e-6 * 1e-3 + 1e-6 * np.abs(p).max(), t     # updates are O(lr)
        p = got.astype(np.float64)   # follow the trainer so the next step checks one update in isolation
    tr.close()


def test_asynchronous_steps_are_the_synchronous_steps():
    """cs_train_step_async (no host synchronisation, loss / MAE summed on the device, alpha passed by value) against
    cs_train_step on a twin trainer: the same weights bit for bit after several steps, Keras's epoch metrics = the mean of the
    per-step scalars, device-side augmentation in the loop, and a reset that clears the sums."""
    import torch
    from cellscreen.augment import ImageDataGenerator
    w = synth.random_cae(seed=12, trivial_bn=True)
    a, b = Trainer(w), Trainer(w)
    gen = ImageDataGenerator.reference()
    X = torch.from_numpy(synth.blob_crops(4, 96)).cuda()
    losses, maes = [], []
    try:
        for i in range(6):
            yb = X[16 * i:16 * i + 16]
            tf = gen.random_transforms(16, (64, 64), np.random.default_rng(i))
            xa = a.augment(yb, tf)
            l, m = a.step(xa, yb, 1e-3)
            losses.append(l); maes.append(m)
            b.step_async(b.augment(yb, tf), yb, 1e-3)
            del xa                                                   # the allocator may hand these bytes out again at once
            junk = torch.full((16, 64, 64), float("nan"), device="cuda")    # ... to this: the steps must have read them before
            del junk
        lo, ma, n = b.read_metrics(reset=True)
        assert n == 6 and abs(lo - np.mean(losses)) <= 1e-6 * abs(lo) and abs(ma - np.mean(maes)) <= 1e-6 * abs(ma)
        pa, ma_ = a.export_flat()
        pb, mb_ = b.export_flat()
        assert np.array_equal(pa, pb) and np.array_equal(ma_, mb_)
        assert b.read_metrics()[2] == 0
        # host batches fall back to a synchronous step and still count
        b.step_async(synth.blob_crops(5, 8), synth.blob_crops(5, 8), 1e-3)
        assert b.read_metrics()[2] == 1
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("shape", ["reference", "generic"])
def test_fit_step_is_gather_augment_and_step_in_one_call(shape):
    """cs_train_fit_step: one library call per fit() batch -- gather train[idx] from the resident training set, the batch's keyed
    augmentation draws (host, counter-based), resample the input only, forward + backward + Adam.  Against the same batch taken
    apart: torch gather -> Trainer.augment with ImageDataGenerator.keyed_transforms -> step_async, on a twin trainer: the same
    weights and moving statistics bit for bit, the same epoch metrics; without augmentation: input == target."""
    import torch
    from cellscreen.augment import ImageDataGenerator
    from cellscreen._lib import CellScreenError
    hw, ch = ((64, 64), spec.CHANNELS) if shape == "reference" else ((64, 128), (8, 16, 32, 32, 16, 8, 1))
    w = synth.random_cae(seed=12, hw=hw, channels=ch, trivial_bn=True)
    a, b, c = Trainer(w), Trainer(w), Trainer(w)
    gen = ImageDataGenerator.reference()
    X = torch.from_numpy(synth.blob_crops(4, 200, hw=hw)).cuda()
    rng = np.random.default_rng(5)
    try:
        for step in range(5):
            idx = rng.permutation(200)[:32].astype(np.int32)
            cfg = gen.config() if step != 3 else None                                   # one step without augmentation
            a.fit_step(X, idx, cfg, seed=77, step=step, lr=1e-3)
            yb = X[torch.from_numpy(idx.astype(np.int64)).cuda()].contiguous()
            xb = b.augment(yb, gen.keyed_transforms(77, step, 32, hw)) if cfg is not None else yb
            b.step_async(xb, yb, 1e-3)
            c.step(xb, yb, 1e-3)                                                        # ... and the synchronous step (one stream sync per step)
        la, ma, na = a.read_metrics()
        lb, mb, nb = b.read_metrics()
        assert na == nb == 5 and la == lb and ma == mb
        pa, mva = a.export_flat()
        pb, mvb = b.export_flat()
        pc, mvc = c.export_flat()
        assert np.array_equal(pa, pb) and np.array_equal(mva, mvb)
        assert np.array_equal(pa, pc) and np.array_equal(mva, mvc)
        with pytest.raises(CellScreenError):
            a.fit_step(X, np.array([0, 200], np.int32), None)                           # outside the training set: refused, nothing enqueued
        with pytest.raises(CellScreenError):
            a.fit_step(X, np.array([-1], np.int32), None)
    finally:
        a.close(); b.close(); c.close()


def test_training_trajectory_tracks_the_oracle():
    """Several full steps.  Adam's early updates are ~lr*sign(g), so fp32 and fp64 trajectories
    separate at the 1e-3 level within a few steps (the numpy oracle in float32 does the same
    against itself in float64): the comparison is loose by necessity."""
    w = synth.random_cae(seed=12, trivial_bn=True)
    x, y = batch(16, seed=3)
    st = T.TrainState(w, dtype=np.float64)
    tr = Trainer(w)
    for step in range(4):
        ref = T.train_step(st, x, y, lr=1e-3)
        loss, _ = tr.step(x, y, lr=1e-3)
        assert abs(loss - ref["loss"]) <= (1e-5 if step == 0 else 5e-3) * ref["loss"], (step, loss, ref["loss"])
    # validation-style evaluation with the moving statistics (fit()'s val_loss)
    l_ref, m_ref = T.evaluate(st, y, y)
    l_got, m_got = tr.evaluate(y, y)
    assert abs(l_got - l_ref) <= 1e-2 * l_ref and abs(m_got - m_ref) <= 1e-2 * m_ref
    # and exactly: the trainer's eval equals the oracle's eval on the trainer's own weights
    pw = tr.weights()
    st2 = T.TrainState(pw, dtype=np.float64)
    l2, m2 = T.evaluate(st2, y, y)
    assert abs(l_got - l2) <= 1e-5 * l2 and abs(m_got - m2) <= 1e-5 * m2
    tr.close()


def test_loss_decreases_and_export_roundtrip():
    w = synth.random_cae(seed=13, trivial_bn=True)
    y = synth.blob_crops(7, 64)
    tr = Trainer(w)
    first = tr.step(y[:32], y[:32])[0]
    for _ in range(30):
        last = tr.step(y[:32], y[:32])[0]
    assert last < 0.7 * first, (first, last)
    p, m = tr.export_flat()
    ev = tr.evaluate(y, y)
    tr2 = Trainer(synth.random_cae(seed=99))
    tr2.load_flat(p, m)
    assert tr2.evaluate(y, y) == ev
    # exported weights drive the screening engine to the same reconstruction error
    from cellscreen.engine import Engine
    e = Engine.from_weights(tr.weights())
    _, mse, _ = e.reconstruct(y, want_recon=False)
    assert abs(float(mse.mean()) - ev[0]) <= 1e-5 * ev[0]
    e.close(); tr.close(); tr2.close()


def test_training_class_mirror(tmp_path):
    """ImprovedAnomalyDetectionTraining with the reference's signatures: split 80/20 seed 42, callbacks, the reference's
    file set (three `.keras` archives + four pickles, CAE_improved_modeltrain.py:271,299-300,437-444), detector, and the
    resulting model_dir loads in the screening class (:480-505 flow) -- from the native files and from the six
    reference files alone."""
    import shutil
    from cellscreen import model_io
    from cellscreen.screening import ProductionMutantScreening
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    cells = synth.blob_crops(21, 640)
    out = str(tmp_path / "models")
    t = ImprovedAnomalyDetectionTraining(out, epochs=3, verbose=0)
    ae0, enc0 = t.create_improved_autoencoder()                                    # (autoencoder, encoder), :184-229
    assert ae0.n_conv == 7 and enc0.n_conv == 3 and enc0.kernels[0] is ae0.kernels[0]
    autoencoder, encoder, history = t.train_autoencoder(cells)
    h = history.history
    assert len(h["loss"]) == 3 and h["loss"][-1] < h["loss"][0] and len(h["val_loss"]) == 3
    assert encoder.n_conv == 3 and autoencoder.n_conv == 7
    for f in ("best_autoencoder.keras", "final_autoencoder.keras", "encoder.keras"):
        assert os.path.exists(os.path.join(out, f)), f
    # Keras 3 restores the best weights at train end: the returned model IS the checkpointed best epoch here
    best = model_io.cae_from_keras(os.path.join(out, "best_autoencoder.keras"))
    final = model_io.cae_from_keras(os.path.join(out, "final_autoencoder.keras"))
    assert all(np.array_equal(a, b) for a, b in zip(final.kernels, autoencoder.kernels))
    assert history.best_epoch == int(np.argmin(h["val_loss"]))
    assert all(np.array_equal(a, b) for a, b in zip(best.kernels, final.kernels))
    mse, mae = t.evaluate_reconstruction_quality(autoencoder, cells)
    assert mse.shape == (640,) and mse.dtype == np.float32
    detectors, scaler, pca = t.create_anomaly_detector(encoder, cells)             # the reference's two-argument call, :394
    assert set(detectors) == {"Conservative", "Moderate"} and pca.n_components_ == 100
    for f in ("scaler.pkl", "pca.pkl", "detector_conservative.pkl", "detector_moderate.pkl", "cae.bin", "detector.bin"):
        assert os.path.exists(os.path.join(out, f)), f
    s = ProductionMutantScreening(out)
    r = s.compute_anomaly_scores(list(cells[:50]))
    # the detector was fit on these cells: scores reproduce sklearn's own decision_function
    feats = s.engine.encode(cells[:50], which=1)
    dec = detectors["Conservative"].decision_function(pca.transform(scaler.transform(feats.copy())))
    assert np.abs(-r["conservative_scores"] - dec).max() <= 1e-4 * np.abs(detectors["Conservative"].dual_coef_).sum()
    assert 0.0 <= r["conservative_anomaly_rate"] <= 0.5
    # the reference's six files alone (improved_detection.py:28-41) are a loadable model_dir: same scores, bit for bit
    six = tmp_path / "six"
    six.mkdir()
    for f in ("best_autoencoder.keras", "encoder.keras", "scaler.pkl", "pca.pkl", "detector_conservative.pkl", "detector_moderate.pkl"):
        shutil.copy(os.path.join(out, f), six / f)
    s2 = ProductionMutantScreening(str(six))
    r2 = s2.compute_anomaly_scores(list(cells[:50]))
    for k in ("reconstruction_mse", "conservative_scores", "moderate_scores", "conservative_predictions"):
        assert np.array_equal(r[k], r2[k]), k


def test_training_class_mirror_on_another_input_shape(tmp_path):
    """`create_improved_autoencoder(input_shape)` is generic in the reference (CAE_improved_modeltrain.py:184).
    128 x 128 crops through the class mirror -- the reference's seven convs on the run-time-shaped trainer (csrc/train_generic.hip):
    two epochs with the reference's augmentation, the reference's files, the detector, and the directory screens 128 x 128 crops with
    the scores scikit-learn's own objects give."""
    from cellscreen import model_io
    from cellscreen.screening import ProductionMutantScreening
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    cells = synth.blob_crops(31, 224, hw=(128, 128))
    out = str(tmp_path / "models128")
    t = ImprovedAnomalyDetectionTraining(out, epochs=2, verbose=0, detector_fit="sklearn")
    ae0, enc0 = t.create_improved_autoencoder((128, 128, 1))
    assert ae0.input_hw == (128, 128) and ae0.channels == spec.CHANNELS and enc0.n_conv == 3
    autoencoder, encoder, history = t.train_autoencoder(cells)
    h = history.history
    assert autoencoder.input_hw == (128, 128) and len(h["loss"]) == 2 and np.isfinite(h["val_loss"]).all() and h["loss"][1] < h["loss"][0]
    assert model_io.cae_from_keras(os.path.join(out, "final_autoencoder.keras")).input_hw == (128, 128)
    mse, _ = t.evaluate_reconstruction_quality(autoencoder, cells)
    assert mse.shape == (224,) and np.isfinite(mse).all()
    detectors, scaler, pca = t.create_anomaly_detector(encoder, cells)
    assert scaler.center_.shape == (16 * 16 * 32,)                                   # the 16 x 16 x 32 bottleneck of a 128 x 128 crop
    s = ProductionMutantScreening(out)
    assert (s.engine.info.height, s.engine.info.width) == (128, 128)
    r = s.compute_anomaly_scores(list(cells[:24]))
    feats = s.engine.encode(cells[:24], which=1)
    dec = detectors["Conservative"].decision_function(pca.transform(scaler.transform(feats.copy())))
    assert np.abs(-r["conservative_scores"] - dec).max() <= 1e-4 * np.abs(detectors["Conservative"].dual_coef_).sum()
    assert np.allclose(r["reconstruction_mse"], mse[:24], rtol=1e-5)


def test_trainer_refuses_shapes_its_kernels_would_get_wrong():
    """the BatchNormalization / pooling kernels the run-time-shaped trainer shares with the reference graph index with
    shifts and masks (powers of two), and the weight-gradient kernel stages rows in LDS: a 96-row input or a 128-wide 128-channel
    conv must be refused at cs_train_create, not trained on silently wrong statistics or fail at the first step."""
    from cellscreen._lib import CellScreenError
    for hw, channels, n_enc, why in (((96, 128), (8, 16, 32, 32, 16, 8, 1), 3, "powers of two"),
                                     ((128, 128), (128, 128, 1), 1, "LDS")):
        w = synth.random_cae(seed=1, hw=hw, channels=channels, n_enc=n_enc, trivial_bn=True)
        with pytest.raises(CellScreenError) as e:
            Trainer(w)
        assert e.value.status == -6 and why in str(e.value), str(e.value)
    # and a shape that IS accepted trains: gradient parity of such a shape is tests/test_gpu_large_variant.py's
    Trainer(synth.random_cae(seed=1, hw=(64, 128), channels=(8, 16, 32, 32, 16, 8, 1), trivial_bn=True)).close()


def test_batch_statistics_of_a_nearly_constant_channel():
    """the forward conv's epilogue takes the BatchNormalization batch statistics in one pass.  A channel whose variance
    is far below mean^2 (here relu(conv) = 50 +- 1e-4: variance ~1e-9 of mean^2) loses its M2 to the rounding of the squares unless
    the sums run about a shift; the BN output computed from the trainer's own relu tensors in float64 says which."""
    w = synth.random_cae(seed=5)
    for l, c in ((1, 7), (3, 11), (4, 3)):
        w.kernels[l][..., c] *= np.float32(2e-5)
        w.biases[l][c] = np.float32(50.0)
    x, y = batch(32, seed=4)
    tr = Trainer(w)
    tr.forward_backward(x, y)
    for l in (1, 3, 4):
        r = tr.tensor(0, l, 32).astype(np.float64)
        a = tr.tensor(1, l, 32).astype(np.float64)
        mu, var = r.mean(axis=(0, 1, 2)), r.var(axis=(0, 1, 2))
        ref = (r - mu) / np.sqrt(var + w.bn_eps) * w.bn_gamma[l].astype(np.float64) + w.bn_beta[l].astype(np.float64)
        if l < 3:
            N, Hh, Ww, Cc = ref.shape
            ref = ref.reshape(N, Hh // 2, 2, Ww // 2, 2, Cc).max(axis=(2, 4))
        c = {1: 7, 3: 11, 4: 3}[l]
        assert var[c] < 1e-6 * mu[c] ** 2, (l, var[c], mu[c])
        err = np.abs(a - ref).max(axis=(0, 1, 2))
        print("layer", l, "var of the flat channel %.3e" % var[c], "BN output err: flat channel %.2e, others %.2e" % (err[c], np.delete(err, c).max()))
        assert err.max() <= 2e-4, (l, int(err.argmax()), err.max())
    tr.close()


def test_trainer_refuses_an_oversized_batch():
    """conv7's loss epilogue leaves 4 bias-gradient partials per cell in a buffer sized once; a batch beyond what it
    (and the 2^31 indexing of the BatchNormalization kernels) holds is refused at the entry of every step call, before anything
    is allocated or launched."""
    from cellscreen._lib import CellScreenError
    import ctypes as C
    tr = Trainer(synth.random_cae(seed=3))
    x = np.zeros((1, 64, 64), np.float32)
    lo, ma = C.c_float(), C.c_float()
    for call in (lambda n: tr._lib.cs_train_step(tr._h, x.ctypes.data, x.ctypes.data, n, 0, 1e-3, C.byref(lo), C.byref(ma)),
                 lambda n: tr._lib.cs_train_step_async(tr._h, x.ctypes.data, x.ctypes.data, n, 0, 1e-3),
                 lambda n: tr._lib.cs_train_forward_backward(tr._h, x.ctypes.data, x.ctypes.data, n, 0, C.byref(lo), C.byref(ma))):
        for n in (8193, 20000, 70000):
            assert call(n) == -6, n        # CS_ERR_UNSUPPORTED, without reading x
    xb, yb = batch(32)
    tr.step(xb, yb)                        # and the trainer still works
    tr.close()


def test_create_anomaly_detector_with_the_reference_signature_alone(tmp_path):
    """create_anomaly_detector(encoder, cell_images) (CAE_improved_modeltrain.py:394) on a fresh instance: no training run, no
    autoencoder argument -- the encoder weight set alone is enough to fit and save the detector."""
    from cellscreen.screening import ProductionMutantScreening
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    w = synth.random_cae(seed=42)
    cells = synth.blob_crops(24, 400)
    out = str(tmp_path / "det_only")
    t = ImprovedAnomalyDetectionTraining(out, verbose=0)
    detectors, scaler, pca = t.create_anomaly_detector(w.encoder_half(), cells)
    assert set(detectors) == {"Conservative", "Moderate"}
    s = ProductionMutantScreening(out)
    r = s.compute_anomaly_scores(list(cells[:40]))
    feats = s.engine.encode(cells[:40], which=1)
    dec = detectors["Moderate"].decision_function(pca.transform(scaler.transform(feats.copy())))
    assert np.abs(-r["moderate_scores"] - dec).max() <= 1e-4 * np.abs(detectors["Moderate"].dual_coef_).sum()
    # with the autoencoder given, the same call also leaves the reference's file set
    out2 = str(tmp_path / "full")
    t2 = ImprovedAnomalyDetectionTraining(out2, verbose=0)
    t2.create_anomaly_detector(w.encoder_half(), cells, autoencoder=w)
    for f in spec.REF_MODEL_FILES:
        assert os.path.exists(os.path.join(out2, f)), f
    r2 = ProductionMutantScreening(out2).compute_anomaly_scores(list(cells[:40]))
    assert np.array_equal(r2["moderate_scores"], r["moderate_scores"]) and r2["reconstruction_mse"].max() < 1.0


def test_training_with_the_reference_augmentation(tmp_path):
    """The default augment="reference": the generator of CAE_improved_modeltrain.py:246-254 on the GPU, input only (:287);
    augment=None is the opt-out."""
    from cellscreen.training import ImprovedAnomalyDetectionTraining
    cells = synth.blob_crops(22, 320)
    t = ImprovedAnomalyDetectionTraining(str(tmp_path / "aug"), epochs=3, verbose=0)
    assert t.augment == "reference"
    _, _, history = t.train_autoencoder(cells)
    h = history.history
    assert len(h["loss"]) == 3 and np.isfinite(h["loss"]).all() and h["loss"][-1] < h["loss"][0]
    t2 = ImprovedAnomalyDetectionTraining(str(tmp_path / "plain"), epochs=3, verbose=0, augment=None)
    _, _, h2 = t2.train_autoencoder(cells)
    assert h2.history["loss"] != h["loss"]                  # the augmented run really saw different inputs


def test_device_inputs_are_ordered_after_torch_work():
    """The handles work on their own non-blocking streams; the wrappers order them after torch's current stream
    (cs_*_wait_stream) instead of relying on the caller to synchronise.  Feed the output of a long chain of async
    torch kernels straight in: the library must see the finished tensor."""
    import torch
    from cellscreen.engine import Engine
    w = synth.random_cae(seed=42)
    e = Engine.from_weights(w)
    base = torch.from_numpy(synth.synth_crops(5, 0, 4096)).cuda()
    want = e.layer_output(base, 1).clone()
    torch.cuda.synchronize()
    for trial in range(5):
        x = torch.zeros_like(base)
        big = torch.randn(4096, 4096, device="cuda")
        for _ in range(20):                                 # keep torch's stream busy for milliseconds
            big = big @ big * 1e-3
        x += base * (1.0 + 0.0 * big[0, 0])                 # x is final only when the chain above has run
        got = e.layer_output(x, 1)
        assert torch.equal(got, want), f"trial {trial}: the library read the tensor before torch finished writing it"
    e.close()


import os  # noqa: E402
