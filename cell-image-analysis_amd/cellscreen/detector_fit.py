"""create_anomaly_detector (CAE_improved_modeltrain.py:394-446): encoder features ->
RobustScaler -> PCA -> two OneClassSVMs.  The FIT runs on the host with scikit-learn, the
same library the reference calls (fitting on device is a "next" row);
the fitted parameters are exported to the arrays the device scoring path consumes."""
from __future__ import annotations

import os
import pickle
from typing import Optional

import numpy as np

from . import spec
from .model_io import detector_params_from_sklearn
from .spec import DetectorParams


def fit_detector(features_flat: np.ndarray, output_dir: Optional[str] = None, pca_random_state=None):
    """features_flat: (N, F) float32 encoder features flattened (h,w,c)  (:401-402).
    Returns (DetectorParams, dict(scaler=..., pca=..., detectors={'Conservative':..,'Moderate':..})).
    When output_dir is given, writes the reference's four pickles (:437-444)."""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    from sklearn.svm import OneClassSVM

    features_flat = np.asarray(features_flat)
    scaler = RobustScaler()                                            # :408
    features_scaled = scaler.fit_transform(features_flat)              # :409
    n_components = min(spec.PCA_MAX_COMPONENTS, features_scaled.shape[1], features_scaled.shape[0] - 1)  # :412
    pca = PCA(n_components=n_components, random_state=pca_random_state)  # :413 (random_state=None there)
    features_reduced = pca.fit_transform(features_scaled)              # :414
    detectors = {                                                      # :420-423
        "Conservative": OneClassSVM(kernel="rbf", gamma="scale", nu=spec.NU_CONSERVATIVE),
        "Moderate": OneClassSVM(kernel="rbf", gamma="scale", nu=spec.NU_MODERATE),
    }
    for det in detectors.values():                                     # :426-427
        det.fit(features_reduced)
    if output_dir is not None:                                         # :437-444
        os.makedirs(output_dir, exist_ok=True)
        for name, obj in (("scaler.pkl", scaler), ("pca.pkl", pca),
                          ("detector_conservative.pkl", detectors["Conservative"]),
                          ("detector_moderate.pkl", detectors["Moderate"])):
            with open(os.path.join(output_dir, name), "wb") as f:
                pickle.dump(obj, f)
    params = detector_params_from_sklearn(scaler, pca, detectors["Conservative"], detectors["Moderate"])
    return params, dict(scaler=scaler, pca=pca, detectors=detectors, features_reduced=features_reduced)
