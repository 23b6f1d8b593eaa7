"""Host-side mirror of the reference's training class (CAE_improved_modeltrain.py:25-446):
same class and method names, argument meaning, hyper-parameters, callback semantics and output
files, with the per-batch arithmetic (fit step, validation pass, reconstruction errors,
encoder features, detector fit) executed by libcellscreen on the GPU.

Out of scope: StarDist cell extraction / dataset assembly (:39-182), plots
and text reports (:304-326, 345-392, 448-478).

* Augmentation (:246-254): the default `augment="reference"` runs the reference's ImageDataGenerator
  settings on the GPU (cellscreen/augment.py + cs_train_augment) -- the reference always trains through
  `datagen.flow` (:287); `augment=None` is the opt-out, any `augment(batch, rng) -> batch` is accepted.
  The reference's quirk that only the INPUT is augmented while the target stays the original image is preserved.
* Callbacks (:263-283): cellscreen/callbacks.py (EarlyStopping, ModelCheckpoint, ReduceLROnPlateau with the Keras
  defaults the reference does not override, e.g. ReduceLROnPlateau's min_delta = 1e-4).
* Files: `best_autoencoder.keras` (ModelCheckpoint, :271), `final_autoencoder.keras`, `encoder.keras` (:299-300) as
  Keras-3 archives (model_io.cae_to_keras), the four pickles (:437-444), and -- after create_anomaly_detector -- the
  native cae.bin / detector.bin beside them, so that both the reference's loader and ProductionMutantScreening
  (output_dir) can read the directory.
* The training set stays resident in HBM; every batch is a device-side gather (no per-batch PCIe traffic).
* `data_parallel=True` (one process per GPU under torch.distributed): every rank draws the SAME shuffled order and
  trains on its 1/W slice of each global batch of `batch_size` (32 cells in total, as the reference's single
  process sees); the flat 337 KB gradient is averaged with one RCCL all-reduce per step (cs_train_forward_backward
  -> all-reduce -> cs_train_apply) so every rank holds identical weights.  `sync_bn=True` (the default) makes the
  BatchNormalization batch statistics those of the WHOLE batch of 32, as the reference's single process computes them
  (CAE_improved_modeltrain.py:192-213 with batch_size=32 at :287): the per-rank {n, mean, M2} triples of a layer are
  all-gathered and merged, and so are the two sums of its backward pass (cellscreen/trainer.py, Trainer.enable_sync_bn);
  `sync_bn=False` keeps per-rank statistics over the rank's slice (what Keras does under data parallelism without
  SyncBatchNormalization) -- a deviation from the reference.  Either way the end of every epoch makes the ranks identical
  before anything is decided: the moving statistics are averaged over the ranks and rank 0's validation loss is what every
  rank's callbacks see, so EarlyStopping and ReduceLROnPlateau fire on the same epoch everywhere.
* Any instance of the layer grammar trains: `create_improved_autoencoder(input_shape)` is generic in the reference
  (:184), and `train_autoencoder` builds the model for the shape of the crops it is given (the reference graph on its tuned
  kernels, other sizes on the run-time-shaped ones of csrc/train_generic.hip).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np

from . import model_io, spec, synth
from .callbacks import FitCallbacks
from .detector_fit import fit_detector, fit_detector_device
from .engine import Engine
from .spec import CAEWeights
from .trainer import Trainer


class History:
    """Stand-in for the Keras History object: .history is the same dict of per-epoch lists."""

    def __init__(self):
        self.history: Dict[str, list] = {"loss": [], "mae": [], "val_loss": [], "val_mae": [], "lr": [], "learning_rate": []}
        self.stopped_epoch: Optional[int] = None
        self.best_epoch: int = -1
        self.lr_reduced_epochs: list = []


class ImprovedAnomalyDetectionTraining:
    def __init__(self, output_dir: str, device_id: int = 0, seed: int = 42, epochs: int = spec.EPOCHS,
                 batch_size: int = spec.BATCH_SIZE, augment="reference", verbose: int = 1,
                 detector_fit: str = "device", data_parallel: bool = False, keras_version: int = 3, sync_bn: bool = True,
                 precision: str = "split16"):
        self.output_dir = output_dir                       # CAE_improved_modeltrain.py:26-27
        os.makedirs(output_dir, exist_ok=True)
        self.device_id = device_id
        self.seed = seed                                   # setup_environment(): seeds 42 (:31-35)
        self.epochs, self.batch_size = epochs, batch_size
        self.augment = augment
        self.verbose = verbose
        if detector_fit not in ("device", "sklearn"):
            raise ValueError("detector_fit must be 'device' (csrc/fit.hip) or 'sklearn' (the reference's library on the host)")
        self.detector_fit = detector_fit
        self.data_parallel = bool(data_parallel)
        self.sync_bn = bool(sync_bn)
        self.precision = precision                         # of the screening engines this class creates (Engine.from_weights)
        self.keras_version = int(keras_version)            # EarlyStopping's restore rule differs (callbacks.py)
        self._autoencoder: Optional[CAEWeights] = None     # what the reference keeps in the Keras objects it returns
        self._best_autoencoder: Optional[CAEWeights] = None

    # ---- model -------------------------------------------------------------------------
    def create_improved_autoencoder(self, input_shape=(64, 64, 1)):
        """:184-229.  Returns (autoencoder, encoder) as the reference does: the initial weight set (Glorot-uniform
        kernels, zero biases, BN gamma 1 / beta 0 / moving mean 0 / moving var 1 -- the Keras defaults) and its encoder
        half, which shares the same arrays as the reference's two Models share their layers.  input_shape is generic as in
        the reference: the same seven convs on another crop size (whether the kernels take that size is decided where the
        weights meet them: Trainer / Engine raise CS_ERR_UNSUPPORTED with the reason)."""
        if len(input_shape) not in (2, 3) or (len(input_shape) == 3 and input_shape[2] != 1):
            raise ValueError(f"input_shape {tuple(input_shape)}: the reference's model takes one grey-level channel")
        ae = synth.random_cae(seed=self.seed, hw=(int(input_shape[0]), int(input_shape[1])), trivial_bn=True)
        return ae, ae.encoder_half()

    # ---- training ------------------------------------------------------------------------
    def _dist(self):
        if not self.data_parallel:
            return None, 0, 1
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("data_parallel=True needs torch.distributed.init_process_group (one process per GPU)")
        return dist, dist.get_rank(), dist.get_world_size()

    def train_autoencoder(self, cell_images):
        """:231-302.  Returns (autoencoder_weights, encoder_weights, history).
        autoencoder_weights / encoder_weights are the in-memory model after fit(): the best-val_loss weights when
        EarlyStopping(restore_best_weights=True) restored them (:264-269; see callbacks.py for when), the last epoch's
        otherwise; best_autoencoder.keras (ModelCheckpoint) is written separately."""
        print("=== Training Autoencoder ===")
        import torch
        from sklearn.model_selection import train_test_split
        X = np.asarray(cell_images).astype("float32")                              # :236-237
        if X.ndim == 4:
            X = X[..., 0]
        X_train, X_val = train_test_split(X, test_size=spec.VAL_SPLIT, random_state=spec.SPLIT_SEED)   # :240
        print(f"Training data: {X_train.shape + (1,)}")
        print(f"Validation data: {X_val.shape + (1,)}")
        dist, rank, world = self._dist()
        if self.batch_size % world:
            raise ValueError(f"batch_size {self.batch_size} does not split over {world} ranks")
        local_b = self.batch_size // world
        dev = torch.device("cuda", self.device_id) if torch.cuda.is_available() else torch.device("cpu")   # no GPU: Trainer() below refuses
        Xd = torch.from_numpy(np.ascontiguousarray(X_train)).to(dev)               # resident: batches are device-side gathers
        Xv = torch.from_numpy(np.ascontiguousarray(X_val)).to(dev)
        ae0, _ = self.create_improved_autoencoder(X.shape[1:3] + (1,))             # :257 (the reference's crops are 64x64)
        tr = Trainer(ae0, device_id=self.device_id)
        grad = None
        if world > 1:
            grad = torch.zeros(tr.n_trainable, dtype=torch.float32, device=dev)
            tr.use_grad_tensor(grad)
            if self.sync_bn:
                if tuple(X.shape[1:3]) == tuple(spec.INPUT_HW) and tuple(ae0.channels) == tuple(spec.CHANNELS):
                    tr.enable_sync_bn(dist, rank, world)
                else:       # cs_train_set_sync_bn serves the reference graph only (CS_ERR_UNSUPPORTED for run-time shapes)
                    import warnings
                    warnings.warn(f"sync_bn is available for the reference 64x64 graph only; training {tuple(X.shape[1:3])} / {tuple(ae0.channels)} "
                                  f"with per-rank BatchNormalization statistics (pass sync_bn=False to silence this)")
        augment = self.augment
        # one GPU, the reference's generator (or none): a fit() batch is ONE library call -- gather, the batch's keyed augmentation
        # draws, forward + backward + Adam (cs_train_fit_step); a caller-supplied hook and the data-parallel step keep the
        # gather -> augment -> step sequence of separate calls
        fit_cfg, use_fit = None, world == 1 and (augment == "reference" or augment is None) and hasattr(tr, "fit_step")
        if augment == "reference":                                                  # datagen of :246-254
            from .augment import ImageDataGenerator, reference_augment
            fit_cfg = ImageDataGenerator.reference().config()
            augment = reference_augment(tr)
        rng = np.random.default_rng(self.seed)                                      # same stream on every rank
        aug_rng = np.random.default_rng(self.seed + 1 + rank)
        steps = len(X_train) // self.batch_size                                     # steps_per_epoch (:288)
        hist = History()
        cb = FitCallbacks(lr=spec.ADAM_LR, keras_version=self.keras_version)
        lr = cb.lr
        best_flat = None
        best_path = os.path.join(self.output_dir, "best_autoencoder.keras")
        try:
            for epoch in range(self.epochs):                                        # epochs=100 (:289)
                perm = rng.permutation(len(X_train))                                # flow(..., shuffle=True)
                tl = tm = 0.0
                if use_fit:
                    idx = np.ascontiguousarray(perm[:steps * self.batch_size], dtype=np.int32).reshape(steps, self.batch_size)
                    for s in range(steps):                                          # the augmentation key: (seed, global step)
                        tr.fit_step(Xd, idx[s], fit_cfg, seed=self.seed + 1, step=epoch * steps + s, lr=lr)
                order = torch.from_numpy(perm).to(dev) if not use_fit else None     # one upload per epoch
                for s in range(steps if not use_fit else 0):
                    lo = s * self.batch_size + rank * local_b
                    yb = Xd[order[lo:lo + local_b]]                                 # gather on the device
                    xb = augment(yb, aug_rng) if augment is not None else yb        # input augmented, target not (:287)
                    if world > 1:
                        l, m = tr.forward_backward(xb, yb)
                        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                        grad /= world
                        tr.apply(lr)                                                # ordered after the all-reduce (Trainer.apply)
                        tl += l; tm += m
                    else:
                        tr.step_async(xb, yb, lr)                                   # no host round trip: the scalars are summed on the device
                if world > 1:
                    # what one process would log: the rank-mean of the training metrics; and ONE set of moving statistics
                    # (identical already under sync_bn; per-rank otherwise), so that every rank validates the same model
                    t = torch.tensor([tl, tm], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    tl, tm = (float(v) / world / max(steps, 1) for v in t.tolist())
                    params, moving = tr.export_flat()
                    mv = torch.from_numpy(moving).to(dev)
                    dist.all_reduce(mv, op=dist.ReduceOp.SUM)
                    tr.load_flat(None, (mv / world).cpu().numpy())
                else:
                    tl, tm, _ = tr.read_metrics(reset=True)                         # Keras's epoch metrics: the mean over the batches
                vl, vm = tr.evaluate(Xv, Xv)                                        # validation_data=(X_val, X_val) (:290)
                if world > 1:
                    # the callbacks act on val_loss: every rank must see the SAME number or the ranks stop / halve the rate on
                    # different epochs (a different stop epoch leaves the others waiting in an all-reduce for ever)
                    v = torch.tensor([vl, vm], dtype=torch.float64, device=dev)
                    dist.broadcast(v, src=0)
                    vl, vm = (float(q) for q in v.tolist())
                for k, v in (("loss", tl), ("mae", tm), ("val_loss", vl), ("val_mae", vm), ("lr", lr), ("learning_rate", lr)):
                    hist.history[k].append(float(v))
                if self.verbose:
                    print(f"Epoch {epoch + 1}/{self.epochs} - loss: {tl:.6f} - mae: {tm:.6f} - val_loss: {vl:.6f} - val_mae: {vm:.6f} - learning_rate: {lr:.2e}")
                act = cb.on_epoch_end(epoch, vl)
                if act.snapshot_best_weights:                                       # EarlyStopping's model.get_weights()
                    best_flat = tr.export_flat()
                if act.save_checkpoint and rank == 0:                               # ModelCheckpoint(save_best_only=True) (:270-275)
                    self._best_autoencoder = tr.weights()
                    model_io.cae_to_keras(best_path, self._best_autoencoder)
                    if self.verbose:
                        print(f"Epoch {epoch + 1}: val_loss improved to {vl:.5f}, saving model to {best_path}")
                if act.lr_reduced and self.verbose:
                    print(f"Epoch {epoch + 1}: ReduceLROnPlateau reducing learning rate to {act.lr}.")
                lr = act.lr
                if act.stop_training:                                               # patience=10
                    if self.verbose:
                        print(f"Epoch {epoch + 1}: early stopping")
                    break
            if cb.restore_best_at_train_end() and best_flat is not None:            # restore_best_weights=True
                if self.verbose:
                    print(f"Restoring model weights from the end of the best epoch: {cb.es_best_epoch + 1}.")
                tr.load_flat(*best_flat)
            final = tr.weights()
        finally:
            tr.close()
        if rank == 0:
            model_io.cae_to_keras(os.path.join(self.output_dir, "final_autoencoder.keras"), final)          # :299
            model_io.cae_to_keras(os.path.join(self.output_dir, "encoder.keras"), final.encoder_half())     # :300
            if self._best_autoencoder is None:             # val_loss was never finite: keep the directory loadable
                self._best_autoencoder = final
                model_io.cae_to_keras(best_path, final)
        hist.stopped_epoch = cb.stopped_epoch if cb.stopped_epoch > 0 else None
        hist.best_epoch = cb.es_best_epoch
        hist.lr_reduced_epochs = list(cb.lr_reduced_epochs)
        self._autoencoder = final
        return final, final.encoder_half(), hist

    # ---- evaluation / detector -----------------------------------------------------------
    def evaluate_reconstruction_quality(self, autoencoder: CAEWeights, cell_images):
        """:328-343 numerics (plots at :345-371 are out of scope) -> (mse_errors, mae_errors)."""
        print("=== Evaluating Reconstruction Quality ===")
        e = Engine.from_weights(autoencoder, device_id=self.device_id, precision=self.precision)
        _, mse, mae = e.reconstruct(np.asarray(cell_images, dtype=np.float32), want_recon=False)
        e.close()
        print(f"MSE - Mean: {np.mean(mse):.6f}, Std: {np.std(mse):.6f}")
        print(f"MAE - Mean: {np.mean(mae):.6f}, Std: {np.std(mae):.6f}")
        return mse, mae

    @staticmethod
    def _padded_autoencoder(encoder: CAEWeights) -> CAEWeights:
        """A full weight set around an encoder-only one (zero decoder): the engine computes encoder features from it."""
        n = spec.N_ENC
        ch = spec.CHANNELS
        if encoder.n_enc != n or tuple(encoder.channels[:n]) != tuple(ch[:n]):
            raise ValueError("an encoder-only weight set of another architecture needs its autoencoder passed alongside")
        ks, bs = list(encoder.kernels), list(encoder.biases)
        g, b, m, v = list(encoder.bn_gamma), list(encoder.bn_beta), list(encoder.bn_mean), list(encoder.bn_var)
        cin = ch[n - 1]
        for l in range(n, len(ch)):
            ks.append(np.zeros((3, 3, cin, ch[l]), np.float32)); bs.append(np.zeros(ch[l], np.float32))
            if l < len(ch) - 1:
                g.append(np.ones(ch[l], np.float32)); b.append(np.zeros(ch[l], np.float32))
                m.append(np.zeros(ch[l], np.float32)); v.append(np.ones(ch[l], np.float32))
            cin = ch[l]
        return CAEWeights(ks, bs, g, b, m, v, encoder.input_hw, n, encoder.bn_eps).validate()

    def create_anomaly_detector(self, encoder: CAEWeights, cell_images, autoencoder: Optional[CAEWeights] = None,
                                best_autoencoder: Optional[CAEWeights] = None):
        """:394-446, the reference's signature `create_anomaly_detector(encoder, cell_images)`.
        encoder.predict and the scaler / PCA / one-class-SVM fit run on the GPU (detector_fit="device", csrc/fit.hip;
        "sklearn" runs the reference's own library on the host instead).  Writes the four pickles (:437-444) AND the
        native cae.bin / detector.bin, so that ProductionMutantScreening(output_dir) loads the directory:
        autoencoder weights = best_autoencoder (what improved_detection.py:28 loads: passed, or kept from
        train_autoencoder, or read back from best_autoencoder.keras), encoder weights = `encoder` (:29)."""
        print("=== Creating Anomaly Detector ===")
        enc_only = encoder if encoder.n_conv == encoder.n_enc else encoder.encoder_half()
        full = autoencoder if autoencoder is not None else (encoder if encoder.n_conv > encoder.n_enc else None)
        if full is None:
            full = self._autoencoder
        if full is None:
            best_path = os.path.join(self.output_dir, "best_autoencoder.keras")
            if os.path.exists(best_path):
                full = model_io.cae_from_keras(best_path)      # what improved_detection.py:28 will load beside the detector
            else:
                print("Warning: no autoencoder given, trained or found in output_dir: the model_dir written here scores with the "
                      "detector only (its reconstruction errors come from a zero decoder)")
                full = self._padded_autoencoder(enc_only)
        e = Engine.from_weights(full, enc_only, device_id=self.device_id, precision=self.precision)
        crops = np.ascontiguousarray(cell_images, dtype=np.float32)
        if crops.ndim == 4:
            crops = crops[..., 0]
        if self.detector_fit == "device":                                               # :408-444
            import torch                                   # plumbing: the features stay on the device between encode and fit
            features_flat = e.encode(torch.from_numpy(crops).cuda(self.device_id), which=1)   # :401-402
            e.close()
            print(f"Flattened features shape: {tuple(features_flat.shape)}")
            params, objs = fit_detector_device(features_flat, output_dir=self.output_dir, device_id=self.device_id)
            del features_flat
        else:
            features_flat = e.encode(crops, which=1)                                    # :401-402
            e.close()
            print(f"Flattened features shape: {features_flat.shape}")
            params, objs = fit_detector(features_flat, output_dir=self.output_dir)
        print(f"PCA reduced to {params.n_components} components")
        print("\nBaseline anomaly rates:")                                              # :430-434
        for name, det in objs["detectors"].items():
            pred = det.predict(objs["features_reduced"])
            print(f"{name}: {np.sum(pred == -1) / len(pred) * 100:.2f}%")
        ae = best_autoencoder if best_autoencoder is not None else self._best_autoencoder
        best_path = os.path.join(self.output_dir, "best_autoencoder.keras")
        if ae is None and os.path.exists(best_path):
            ae = model_io.cae_from_keras(best_path)
        if ae is None:
            ae = full
        for name, wset in (("best_autoencoder.keras", ae), ("encoder.keras", enc_only)):   # the six files of improved_detection.py:28-41
            if not os.path.exists(os.path.join(self.output_dir, name)):
                model_io.cae_to_keras(os.path.join(self.output_dir, name), wset)
        model_io.save_model_dir(self.output_dir, ae, enc_only, params)                     # written last: newer than the files it mirrors
        return objs["detectors"], objs["scaler"], objs["pca"]
