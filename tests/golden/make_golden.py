"""Generates the committed golden fixtures.  Run in the build container:
    python tests/golden/make_golden.py
The reference has no tests, fixtures or model files (its .gitignore:105-129 excludes them)
and TensorFlow/Keras is not installed, so these vectors come from
  * an independent implementation of the Keras graph (torch-CPU functional ops, float64), and
  * the real scikit-learn 1.7.2 objects the reference calls (RobustScaler, PCA, OneClassSVM),
never from the oracle or from the HIP path they are used to check.

golden_cae.npz       weights, 8 crops, expected features / recon / mse / mae / per-layer stats
golden_detector.npz  fitted sklearn parameters, 32 test feature rows, expected scaled / pca /
                     decision_function / predict of both detectors
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd"))

from cellscreen import spec, synth  # noqa: E402
from cellscreen.detector_fit import fit_detector  # noqa: E402


def torch_forward(w, x, dtype):
    """Keras graph of CAE_improved_modeltrain.py:191-216 with torch functional ops (NCHW)."""
    import torch
    import torch.nn.functional as F
    T = lambda a: torch.from_numpy(np.asarray(a)).to(dtype)
    h = T(x)[:, None]
    layers = []
    feats = None
    for l in range(w.n_conv):
        if l > w.n_enc:
            h = F.interpolate(h, scale_factor=2, mode="nearest")       # UpSampling2D((2,2))
        h = F.conv2d(h, T(w.kernels[l]).permute(3, 2, 0, 1), T(w.biases[l]), padding=1)  # Conv2D 3x3 same
        if l < w.n_conv - 1:
            h = F.relu(h)                                               # activation='relu'
            h = F.batch_norm(h, T(w.bn_mean[l]), T(w.bn_var[l]), T(w.bn_gamma[l]), T(w.bn_beta[l]),
                             False, 0.0, w.bn_eps)                      # BatchNormalization, inference
            if l < w.n_enc:
                h = F.max_pool2d(h, 2)                                  # MaxPooling2D((2,2))
        else:
            h = torch.sigmoid(h)                                        # activation='sigmoid'
        layers.append(h.permute(0, 2, 3, 1).contiguous().numpy())       # NHWC
        if l == w.n_enc - 1:
            feats = layers[-1].reshape(len(x), -1)                      # (h,w,c) flatten
    return layers, feats


def main():
    import torch
    torch.set_num_threads(4)
    w = synth.random_cae(seed=42)
    x = synth.synth_crops(seed=42, first_cell=0, n=8)
    # a couple of structured crops so activations are not all noise-like
    x[6:8] = synth.blob_crops(seed=3, n=2)
    layers, feats = torch_forward(w, x, torch.float64)
    recon = layers[-1][..., 0]
    d = x.astype(np.float64) - recon
    out = dict(crops=x, features=feats.astype(np.float64), recon=recon.astype(np.float64),
               mse=np.mean(d * d, axis=(1, 2)), mae=np.mean(np.abs(d), axis=(1, 2)),
               bn_eps=np.float32(w.bn_eps))
    for l in range(w.n_conv):
        out[f"conv{l}_kernel"] = w.kernels[l]
        out[f"conv{l}_bias"] = w.biases[l]
        flat = layers[l].reshape(len(x), -1)
        out[f"layer{l}_sum"] = flat.sum(axis=1)
        out[f"layer{l}_sumsq"] = (flat * flat).sum(axis=1)
        out[f"layer{l}_first64"] = flat[:, :64].copy()
    for l in range(w.n_conv - 1):
        out[f"bn{l}_gamma"], out[f"bn{l}_beta"] = w.bn_gamma[l], w.bn_beta[l]
        out[f"bn{l}_mean"], out[f"bn{l}_var"] = w.bn_mean[l], w.bn_var[l]
    np.savez_compressed(os.path.join(HERE, "golden_cae.npz"), **out)

    # ---- detector: fit with the real sklearn on torch-float32 features of 300 synthetic crops
    xt = np.concatenate([synth.synth_crops(42, 1000, 200), synth.blob_crops(5, 100)])
    _, ftrain = torch_forward(w, xt, torch.float32)
    params, objs = fit_detector(ftrain.astype(np.float32), pca_random_state=0)
    xq = np.concatenate([synth.synth_crops(42, 5000, 24), synth.blob_crops(9, 8)])
    _, fq = torch_forward(w, xq, torch.float32)
    fq = fq.astype(np.float32)
    scaled = objs["scaler"].transform(fq.copy())           # improved_detection.py:134
    pca = objs["pca"].transform(scaled)                    # :135
    dc, dm = objs["detectors"]["Conservative"], objs["detectors"]["Moderate"]
    det = dict(test_features=fq, scaled=scaled, pca=pca,
               cons_dec=dc.decision_function(pca), mod_dec=dm.decision_function(pca),       # :141-142
               cons_pred=dc.predict(pca).astype(np.int64), mod_pred=dm.predict(pca).astype(np.int64),  # :138-139
               scaler_center=params.scaler_center, scaler_scale=params.scaler_scale,
               pca_components=params.pca_components, pca_mean=params.pca_mean, pca_mean_proj=params.pca_mean_proj,
               cons_sv=params.conservative.support_vectors, cons_coef=params.conservative.dual_coef,
               cons_gamma=np.float64(params.conservative.gamma), cons_rho=np.float64(params.conservative.rho),
               mod_sv=params.moderate.support_vectors, mod_coef=params.moderate.dual_coef,
               mod_gamma=np.float64(params.moderate.gamma), mod_rho=np.float64(params.moderate.rho))
    np.savez_compressed(os.path.join(HERE, "golden_detector.npz"), **det)
    for f in ("golden_cae.npz", "golden_detector.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
    print("dtypes:", scaled.dtype, pca.dtype, det["cons_dec"].dtype, "n_sv", params.conservative.n_sv, params.moderate.n_sv,
          "n_comp", params.n_components)


if __name__ == "__main__":
    main()
