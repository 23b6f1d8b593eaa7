"""One-off full-size parity check of the device one-class-SVM fit (csrc/fit.hip) against libsvm at the reference's
training-set size: 50,000 cells' encoder features -> device scaler / PCA / projection -> the SAME 50,000 x 100 float64
matrix is handed to scikit-learn's OneClassSVM (about a minute of host time) and to cs_fit_ocsvm.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cell-image-analysis_amd"))
from cellscreen import detector_fit as df, synth  # noqa: E402
from cellscreen.engine import Engine  # noqa: E402
from sklearn.svm import OneClassSVM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
e = Engine.from_weights(synth.random_cae(seed=42), None, None)
feats = np.concatenate([e.encode(synth.blob_crops(100 + i, min(10000, n - i)), which=0) for i in range(0, n, 10000)])
e.close()
det, objs = df.fit_detector_device(feats)
x = objs["features_reduced"].astype(np.float64)
out = {"n": n, "n_components": int(x.shape[1])}
for name, nu in (("Conservative", 0.05), ("Moderate", 0.10)):
    r = objs["solved"][name]
    t0 = time.perf_counter()
    sk = OneClassSVM(kernel="rbf", gamma="scale", nu=nu).fit(x)
    dt = time.perf_counter() - t0
    alpha = np.zeros(n)
    alpha[sk.support_] = sk.dual_coef_.ravel()
    sv_dev = np.flatnonzero(r["alpha"] > 0)
    out[name] = {"libsvm_s": round(dt, 2), "n_iter": [int(sk.n_iter_), int(r["n_iter"])],
                 "n_sv": [int(sk.support_.size), int(sv_dev.size)],
                 "support_sets_equal": bool(np.array_equal(sv_dev, sk.support_)),
                 "max_abs_dalpha": float(np.abs(alpha - r["alpha"]).max()),
                 "rho": [float(-sk.intercept_[0]), float(r["rho"])],
                 "gamma_equal": bool(sk._gamma == objs["detectors"][name]._gamma)}
print(json.dumps(out))
