"""Detector fit (CAE_improved_modeltrain.py:408-427) at the reference's scale: N training cells' encoder features
(N x 2048 float32) -> RobustScaler, PCA(100), two one-class SVMs.  Times the device fit (csrc/fit.hip) and, with
--sklearn, the host scikit-learn fit the reference runs; prints one JSON line.

    python tools/bench_fit.py --n 50000 --sklearn
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cell-image-analysis_amd"))
from cellscreen import detector_fit as df, synth  # noqa: E402
from cellscreen.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)          # BASELINE.json configs[1]: 50,000 training crops
    ap.add_argument("--sklearn", action="store_true", help="also time the host scikit-learn fit (about 1.5 min at 50k)")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import torch
    w = synth.random_cae(seed=42)
    e = Engine.from_weights(w, None, None)
    feats = np.concatenate([e.encode(synth.blob_crops(100 + i, min(10000, a.n - i)), which=0) for i in range(0, a.n, 10000)])
    e.close()
    feats_dev = torch.from_numpy(feats).cuda()
    out = {"n": a.n, "n_features": int(feats.shape[1])}
    for r in range(a.repeat):                                  # first pass pays the allocations
        t = {}
        t0 = time.perf_counter()
        det, objs = df.fit_detector_device(feats_dev, timings=t)
        t["total_s"] = time.perf_counter() - t0
        out[f"device_pass{r}"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}
    out["n_sv"] = [det.conservative.n_sv, det.moderate.n_sv]
    if a.sklearn:
        from sklearn.decomposition import PCA
        from sklearn.preprocessing import RobustScaler
        from sklearn.svm import OneClassSVM
        t = {}
        t0 = time.perf_counter(); sc = RobustScaler(); xs = sc.fit_transform(feats); t["scaler_s"] = time.perf_counter() - t0
        t1 = time.perf_counter(); p = PCA(n_components=100, random_state=0); red = p.fit_transform(xs); t["pca_s"] = time.perf_counter() - t1
        iters = []
        for name, nu in (("conservative", 0.05), ("moderate", 0.10)):
            t1 = time.perf_counter(); d = OneClassSVM(kernel="rbf", gamma="scale", nu=nu).fit(red)
            t[f"svm_{name}_s"] = time.perf_counter() - t1; iters.append(int(d.n_iter_))
        t["total_s"] = time.perf_counter() - t0
        out["sklearn"] = {k: round(v, 3) for k, v in t.items()}
        out["sklearn"]["svm_iters"] = iters
        out["sklearn"]["cores"] = os.cpu_count()
        out["scaler_equal"] = bool(np.array_equal(det.scaler_center, sc.center_) and np.array_equal(det.scaler_scale, sc.scale_))
        out["speedup"] = round(out["sklearn"]["total_s"] / out[f"device_pass{a.repeat - 1}"]["total_s"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
