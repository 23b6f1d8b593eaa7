"""Host-side mirror of the reference's training class (CAE_improved_modeltrain.py:25-446):
same class and method names, argument meaning, hyper-parameters, callback semantics and output
file roles, with the per-batch arithmetic (fit step, validation pass, reconstruction errors,
encoder features) executed by libcellscreen on the GPU.

Out of scope: StarDist cell extraction / dataset assembly (:39-182), plots
and text reports (:304-326, 345-392, 448-478).  Augmentation (:246-254): `augment="reference"` runs the
reference's ImageDataGenerator settings on the GPU (cellscreen/augment.py + cs_train_augment), or pass
any `augment(batch, rng) -> batch`; the reference's quirk that only the INPUT is augmented while the
target stays the original image (:287) is preserved.  Model files are written in the native
format (model_io.save_model_dir) under the reference's roles: best (ModelCheckpoint, :270-275),
final and encoder (:299-300), scaler/pca/detectors (:437-444, also as the reference's pickles)."""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np

from . import model_io, spec, synth
from .detector_fit import fit_detector, fit_detector_device
from .engine import Engine
from .spec import CAEWeights
from .trainer import Trainer


class History:
    """Stand-in for the Keras History object: .history is the same dict of per-epoch lists."""

    def __init__(self):
        self.history: Dict[str, list] = {"loss": [], "mae": [], "val_loss": [], "val_mae": [], "lr": []}


class ImprovedAnomalyDetectionTraining:
    def __init__(self, output_dir: str, device_id: int = 0, seed: int = 42, epochs: int = spec.EPOCHS,
                 batch_size: int = spec.BATCH_SIZE, augment: Optional[Callable] = None, verbose: int = 1,
                 detector_fit: str = "device"):
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

    # ---- model -------------------------------------------------------------------------
    def create_improved_autoencoder(self, input_shape=(64, 64, 1)) -> CAEWeights:
        """:184-229.  Returns the initial weight set (Glorot-uniform kernels, zero biases, BN
        gamma 1 / beta 0 / moving mean 0 / moving var 1 -- the Keras defaults); the autoencoder
        and the encoder of the reference share these layers."""
        if tuple(input_shape[:2]) != spec.INPUT_HW:
            raise NotImplementedError("this build has kernels for 64x64 crops")
        return synth.random_cae(seed=self.seed, trivial_bn=True)

    # ---- training ------------------------------------------------------------------------
    def train_autoencoder(self, cell_images):
        """:231-302.  Returns (autoencoder_weights, encoder_weights, history).
        autoencoder_weights / encoder_weights are the in-memory model after fit() -- i.e. the
        best-val_loss weights when EarlyStopping fired with restore_best_weights=True (:264-269),
        the last epoch's otherwise; best_autoencoder (ModelCheckpoint) is kept separately."""
        print("=== Training Autoencoder ===")
        from sklearn.model_selection import train_test_split
        X = np.asarray(cell_images).astype("float32")                              # :236-237
        if X.ndim == 4:
            X = X[..., 0]
        X_train, X_val = train_test_split(X, test_size=spec.VAL_SPLIT, random_state=spec.SPLIT_SEED)   # :240
        print(f"Training data: {X_train.shape + (1,)}")
        print(f"Validation data: {X_val.shape + (1,)}")
        tr = Trainer(self.create_improved_autoencoder(), device_id=self.device_id)
        if self.augment == "reference":                                             # datagen of :246-254
            from .augment import reference_augment
            self.augment = reference_augment(tr)
        rng = np.random.default_rng(self.seed)
        steps = len(X_train) // self.batch_size                                     # steps_per_epoch (:288)
        hist = History()
        lr = spec.ADAM_LR
        best_val, best_flat, best_epoch = np.inf, None, -1
        es_wait, rl_wait = 0, 0
        stopped_epoch = None
        for epoch in range(self.epochs):                                            # epochs=100 (:289)
            order = rng.permutation(len(X_train))                                   # flow(..., shuffle=True)
            tl = tm = 0.0
            for s in range(steps):
                idx = order[s * self.batch_size:(s + 1) * self.batch_size]
                yb = X_train[idx]
                xb = self.augment(yb, rng) if self.augment is not None else yb      # input augmented, target not (:287)
                l, m = tr.step(xb, yb, lr)
                tl += l; tm += m
            vl, vm = tr.evaluate(X_val, X_val)                                      # validation_data=(X_val, X_val) (:290)
            for k, v in (("loss", tl / max(steps, 1)), ("mae", tm / max(steps, 1)), ("val_loss", vl), ("val_mae", vm), ("lr", lr)):
                hist.history[k].append(float(v))
            if self.verbose:
                print(f"Epoch {epoch + 1}/{self.epochs} - loss: {tl / max(steps, 1):.6f} - mae: {tm / max(steps, 1):.6f} - val_loss: {vl:.6f} - val_mae: {vm:.6f} - lr: {lr:.2e}")
            # ModelCheckpoint(monitor='val_loss', save_best_only=True) (:270-275) and the weights
            # EarlyStopping(restore_best_weights=True) would restore (:264-269)
            if vl < best_val:
                best_val, best_epoch = vl, epoch
                best_flat = tr.export_flat()
                model_io.save_model_dir(os.path.join(self.output_dir, "best_autoencoder"), tr.weights())
                es_wait = 0
            else:
                es_wait += 1
            # ReduceLROnPlateau(factor=0.5, patience=5, min_lr=1e-6) (:276-282)
            if vl < getattr(self, "_rl_best", np.inf):
                self._rl_best, rl_wait = vl, 0
            else:
                rl_wait += 1
                if rl_wait >= spec.RLROP_PATIENCE:
                    new_lr = max(lr * spec.RLROP_FACTOR, spec.RLROP_MIN_LR)
                    if new_lr < lr and self.verbose:
                        print(f"Epoch {epoch + 1}: ReduceLROnPlateau reducing learning rate to {new_lr}.")
                    lr, rl_wait = new_lr, 0
            if es_wait >= spec.ES_PATIENCE:                                          # patience=10
                stopped_epoch = epoch
                if self.verbose:
                    print(f"Epoch {epoch + 1}: early stopping; restoring best weights from epoch {best_epoch + 1}")
                tr.load_flat(*best_flat)                                             # restore_best_weights=True
                break
        self._rl_best = np.inf
        final = tr.weights()
        model_io.save_model_dir(os.path.join(self.output_dir, "final_autoencoder"), final)            # :299
        model_io.save_model_dir(os.path.join(self.output_dir, "encoder"), final)                      # :300 (encoder half is read from it)
        hist.stopped_epoch, hist.best_epoch = stopped_epoch, best_epoch
        tr.close()
        return final, final.encoder_half(), hist

    # ---- evaluation / detector -----------------------------------------------------------
    def evaluate_reconstruction_quality(self, autoencoder: CAEWeights, cell_images):
        """:328-343 numerics (plots at :345-371 are out of scope) -> (mse_errors, mae_errors)."""
        print("=== Evaluating Reconstruction Quality ===")
        e = Engine.from_weights(autoencoder, device_id=self.device_id)
        _, mse, mae = e.reconstruct(np.asarray(cell_images, dtype=np.float32), want_recon=False)
        e.close()
        print(f"MSE - Mean: {np.mean(mse):.6f}, Std: {np.std(mse):.6f}")
        print(f"MAE - Mean: {np.mean(mae):.6f}, Std: {np.std(mae):.6f}")
        return mse, mae

    def create_anomaly_detector(self, encoder: CAEWeights, cell_images, autoencoder: Optional[CAEWeights] = None,
                                best_autoencoder: Optional[CAEWeights] = None):
        """:394-446.  encoder.predict and the scaler / PCA / one-class-SVM fit on the GPU
        (detector_fit="device", csrc/fit.hip; "sklearn" runs the reference's own library on the
        host instead), the four pickles AND a native model_dir at output_dir that
        ProductionMutantScreening(output_dir) loads: autoencoder weights = best_autoencoder
        (what improved_detection.py:28 loads), encoder weights = `encoder` (:29)."""
        print("=== Creating Anomaly Detector ===")
        full = autoencoder if autoencoder is not None else None
        if full is None:
            raise ValueError("pass autoencoder= (the full weight set whose encoder half is `encoder`)")
        e = Engine.from_weights(full, encoder if encoder.n_conv == encoder.n_enc else None, device_id=self.device_id)
        crops = np.ascontiguousarray(cell_images, dtype=np.float32)
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
        ae = best_autoencoder if best_autoencoder is not None else full
        model_io.save_model_dir(self.output_dir, ae, encoder if encoder.n_conv == encoder.n_enc else None, params)
        return objs["detectors"], objs["scaler"], objs["pca"]
