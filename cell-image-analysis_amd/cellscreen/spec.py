"""Constants and parameter containers of the reference path, each with the reference line
it restates.  Nothing here computes on cells; it is shared by the host mirror, the model
files and the test oracle wrapper.

Reference graph: CAE_improved_modeltrain.py:184-229 (create_improved_autoencoder).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

# CAE_improved_modeltrain.py:184 input_shape=(64, 64, 1)
INPUT_HW: Tuple[int, int] = (64, 64)
# filters of the seven Conv2D layers, :191,195,199 (encoder) :204,208,212,216 (decoder)
CHANNELS: Tuple[int, ...] = (32, 64, 32, 32, 64, 32, 1)
N_ENC = 3               # :191-201  three conv + BN + MaxPooling2D blocks
BN_EPS = 1e-3           # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99      # Keras default momentum (training)
# Adam(learning_rate=0.001), Keras defaults beta_1=0.9 beta_2=0.999 epsilon=1e-7   :223-227
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 1e-3, 0.9, 0.999, 1e-7
BATCH_SIZE = 32         # :287
EPOCHS = 100            # :289
VAL_SPLIT, SPLIT_SEED = 0.2, 42                         # :240
ES_PATIENCE = 10        # EarlyStopping(patience=10, restore_best_weights=True) :264-269
RLROP_FACTOR, RLROP_PATIENCE, RLROP_MIN_LR = 0.5, 5, 1e-6   # :276-282
RLROP_MIN_DELTA = 1e-4  # ReduceLROnPlateau's Keras default min_delta (not passed at :276-282): improvement iff val < best - 1e-4
PCA_MAX_COMPONENTS = 100                                 # :412
NU_CONSERVATIVE, NU_MODERATE = 0.05, 0.10                # :421-422
MIN_TRAINING_CELLS = 500                                 # :491-493

# files of a reference model_dir, improved_detection.py:28-41
REF_MODEL_FILES = ("best_autoencoder.keras", "encoder.keras", "scaler.pkl", "pca.pkl",
                   "detector_conservative.pkl", "detector_moderate.pkl")
# native files written/read by this package (see DESIGN.md "model_dir")
NATIVE_CAE, NATIVE_DETECTOR, NATIVE_MANIFEST = "cae.bin", "detector.bin", "manifest.json"

# result keys of compute_anomaly_scores, improved_detection.py:144-153, in order
SCORE_KEYS = ("reconstruction_mse", "reconstruction_mae", "conservative_predictions",
              "moderate_predictions", "conservative_scores", "moderate_scores",
              "conservative_anomaly_rate", "moderate_anomaly_rate")
# CSV schemas, improved_detection.py:202-212 and :225-234
SUMMARY_COLUMNS = ("sample_name", "total_cells", "files_processed", "conservative_anomaly_rate",
                   "moderate_anomaly_rate", "mean_mse", "std_mse", "mean_mae", "std_mae")
DETAIL_COLUMNS = ("sample_name", "cell_id", "mse", "mae", "conservative_anomaly",
                  "moderate_anomaly", "conservative_score", "moderate_score")


def layer_table(hw=INPUT_HW, channels=CHANNELS, n_enc=N_ENC):
    """Per conv: dict(cin, cout, conv_hw, out_hw, pool, ups, macs) -- SURVEY.md Appendix A.1."""
    h, w = hw
    rows = []
    cin = 1
    for l, cout in enumerate(channels):
        is_enc = l < n_enc
        ups = (not is_enc) and l > n_enc          # UpSampling2D precedes every decoder conv but the first
        ch, cw = (h * 2, w * 2) if ups else (h, w)
        oh, ow = (ch // 2, cw // 2) if is_enc else (ch, cw)
        rows.append(dict(cin=cin, cout=cout, conv_hw=(ch, cw), out_hw=(oh, ow), pool=is_enc, ups=ups,
                         macs=ch * cw * 9 * cin * cout))
        h, w, cin = oh, ow, cout
    return rows


@dataclass
class CAEWeights:
    """One Keras weight set: kernels HWIO (3,3,cin,cout) float32, BN per conv but the last."""
    kernels: List[np.ndarray]
    biases: List[np.ndarray]
    bn_gamma: List[np.ndarray]
    bn_beta: List[np.ndarray]
    bn_mean: List[np.ndarray]
    bn_var: List[np.ndarray]
    input_hw: Tuple[int, int] = INPUT_HW
    n_enc: int = N_ENC
    bn_eps: float = BN_EPS

    @property
    def n_conv(self) -> int:
        return len(self.kernels)

    @property
    def channels(self) -> Tuple[int, ...]:
        return tuple(int(k.shape[3]) for k in self.kernels)

    def n_params(self) -> int:
        n = sum(k.size + b.size for k, b in zip(self.kernels, self.biases))
        return n + sum(4 * g.size for g in self.bn_gamma)

    def bn_scale_shift(self):
        """Inference BN as y = x*s + t, computed in float32 exactly as csrc/api.hip does:
        s = gamma / sqrt(var + eps), t = beta - mean * s."""
        eps = np.float32(self.bn_eps)
        s = [(g / np.sqrt(v + eps)).astype(np.float32) for g, v in zip(self.bn_gamma, self.bn_var)]
        t = [(b - m * si).astype(np.float32) for b, m, si in zip(self.bn_beta, self.bn_mean, s)]
        return s, t

    def encoder_half(self) -> "CAEWeights":
        n = self.n_enc
        return CAEWeights(self.kernels[:n], self.biases[:n], self.bn_gamma[:n], self.bn_beta[:n],
                          self.bn_mean[:n], self.bn_var[:n], self.input_hw, n, self.bn_eps)

    def validate(self):
        cin = 1
        for l, (k, b) in enumerate(zip(self.kernels, self.biases)):
            if k.dtype != np.float32 or k.ndim != 4 or k.shape[:3] != (3, 3, cin):
                raise ValueError(f"conv {l}: kernel must be float32 (3,3,{cin},cout), got {k.dtype} {k.shape}")
            if b.shape != (k.shape[3],):
                raise ValueError(f"conv {l}: bias shape {b.shape}")
            cin = k.shape[3]
        nbn = len(self.bn_gamma)
        if not (len(self.bn_beta) == len(self.bn_mean) == len(self.bn_var) == nbn):
            raise ValueError("BN lists differ in length")
        return self


@dataclass
class OCSVMParams:
    support_vectors: np.ndarray   # (n_sv, n_components) float64
    dual_coef: np.ndarray         # (n_sv,) float64
    gamma: float
    rho: float                    # = -intercept_ = offset_

    @property
    def n_sv(self) -> int:
        return int(self.support_vectors.shape[0])


@dataclass
class DetectorParams:
    """RobustScaler + PCA + the two OneClassSVMs of CAE_improved_modeltrain.py:408-427."""
    scaler_center: np.ndarray     # (F,) float32   center_
    scaler_scale: np.ndarray      # (F,) float64   scale_
    pca_components: np.ndarray    # (C,F) float32  components_
    pca_mean: np.ndarray          # (F,) float32   mean_
    pca_mean_proj: np.ndarray     # (C,) float32   mean_.reshape(1,-1) @ components_.T
    conservative: OCSVMParams = field(default=None)
    moderate: OCSVMParams = field(default=None)

    @property
    def n_features(self) -> int:
        return int(self.pca_components.shape[1])

    @property
    def n_components(self) -> int:
        return int(self.pca_components.shape[0])
