"""Pins the CPU oracle (oracle/cae_oracle.c) -- the checker of every GPU parity test --
against (a) the committed golden vectors (torch-float64 graph + real sklearn outputs) and
(b) live torch / sklearn runs on fresh random inputs.  No GPU needed."""
import numpy as np
import pytest

import helpers as H
from cellscreen import spec, synth
from oracle import oracle


def test_synth_hash_numpy_equals_c():
    for seed, first, n in ((42, 0, 5), (7, 123456789, 3), (2**40 + 3, 2**33, 2)):
        assert np.array_equal(synth.synth_crops(seed, first, n), oracle.synth_crops(seed, first, n))
    x = oracle.synth_crops(42, 0, 64)
    assert x.dtype == np.float32 and x.min() >= 0.0 and x.max() < 1.0
    assert abs(float(x.mean()) - 0.5) < 0.01


def test_layer_table_matches_survey():
    rows = spec.layer_table()
    assert [r["macs"] for r in rows] == [1179648, 18874368, 4718592, 589824, 4718592, 18874368, 1179648]
    assert sum(r["macs"] for r in rows) == 50135040
    assert sum(r["macs"] for r in rows[:3]) == 24772608
    assert rows[2]["out_hw"] == (8, 8) and rows[2]["cout"] == 32
    assert synth.random_cae().n_params() == 84801


def test_oracle_vs_golden_cae(golden_cae):
    g = golden_cae
    w = H.cae_from_golden(g)
    for acc64 in (True, False):
        r = oracle.cae_forward(w, g["crops"], acc64=acc64, layers=True)
        H.assert_close_scaled(r["features"], g["features"], H.TOL_FEATURES, "features")
        assert np.abs(r["recon"] - g["recon"]).max() <= H.TOL_RECON
        H.assert_rel(r["mse"], g["mse"], H.TOL_ERR_REL, "mse")
        H.assert_rel(r["mae"], g["mae"], H.TOL_ERR_REL, "mae")
        for l in range(7):
            flat = r["layers"][l].reshape(len(g["crops"]), -1).astype(np.float64)
            H.assert_close_scaled(flat[:, :64], g[f"layer{l}_first64"], 1e-5, f"layer{l} head")
            H.assert_rel(flat.sum(axis=1), g[f"layer{l}_sum"], 1e-5, f"layer{l} sum")
            H.assert_rel((flat * flat).sum(axis=1), g[f"layer{l}_sumsq"], 1e-5, f"layer{l} sumsq")


def test_oracle_vs_golden_detector(golden_det):
    g = golden_det
    det = H.det_from_golden(g)
    scaled, pca = oracle.scaler_pca(det, g["test_features"], acc64=True)
    # the scaler is elementwise: bit-exact against sklearn's float32 output
    assert np.array_equal(scaled, g["scaled"])
    H.assert_close_scaled(pca, g["pca"], H.TOL_STAGE, "pca")
    for name in ("cons", "mod"):
        p = det.conservative if name == "cons" else det.moderate
        dec, pred = oracle.ocsvm_decision(p, g["pca"])
        tol = H.TOL_DEC_STAGE * np.abs(p.dual_coef).sum()
        assert np.abs(dec - g[f"{name}_dec"]).max() <= tol
        H.flags_agree(dec, pred, g[f"{name}_dec"], g[f"{name}_pred"], tol, name)


def test_oracle_vs_live_torch_random_weights():
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    w = synth.random_cae(seed=123)
    x = np.concatenate([synth.synth_crops(9, 77, 3), synth.blob_crops(1, 2)])
    r = oracle.cae_forward(w, x, acc64=True, layers=True)
    T = lambda a: torch.from_numpy(np.asarray(a)).double()
    h = T(x)[:, None]
    for l in range(7):
        if l > 3:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
        h = F.conv2d(h, T(w.kernels[l]).permute(3, 2, 0, 1), T(w.biases[l]), padding=1)
        if l < 6:
            h = F.batch_norm(F.relu(h), T(w.bn_mean[l]), T(w.bn_var[l]), T(w.bn_gamma[l]), T(w.bn_beta[l]), False, 0.0, w.bn_eps)
            if l < 3:
                h = F.max_pool2d(h, 2)
        else:
            h = torch.sigmoid(h)
        H.assert_close_scaled(r["layers"][l], h.permute(0, 2, 3, 1).numpy(), 2e-6, f"layer {l}")


def test_oracle_negative_bn_scale_pool_order():
    """BN sits between ReLU and max-pool (CAE...:191-193): with a negative gamma the pool must
    see the BN output, not the ReLU output."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    w = synth.random_cae(seed=5)
    for l in range(6):
        w.bn_gamma[l][::2] *= -1.0
    x = synth.synth_crops(3, 0, 2)
    r = oracle.cae_forward(w, x, acc64=True, layers=True)
    T = lambda a: torch.from_numpy(np.asarray(a)).double()
    h = F.conv2d(T(x)[:, None], T(w.kernels[0]).permute(3, 2, 0, 1), T(w.biases[0]), padding=1)
    h = F.batch_norm(F.relu(h), T(w.bn_mean[0]), T(w.bn_var[0]), T(w.bn_gamma[0]), T(w.bn_beta[0]), False, 0.0, w.bn_eps)
    h = F.max_pool2d(h, 2)
    H.assert_close_scaled(r["layers"][0], h.permute(0, 2, 3, 1).numpy(), 2e-6, "layer 0 with negative gamma")


def test_oracle_vs_live_sklearn():
    pytest.importorskip("sklearn")
    from cellscreen.detector_fit import fit_detector
    rng = np.random.default_rng(0)
    feats = (rng.standard_normal((260, 2048)) * rng.uniform(0.1, 2.0, 2048) + rng.uniform(-1, 1, 2048)).astype(np.float32)
    feats[:, 5] = 0.25   # zero IQR -> scale_ 1.0 (sklearn _data.py:1672-1677)
    params, objs = fit_detector(feats, pca_random_state=0)
    assert params.n_components == 100 and params.scaler_scale.dtype == np.float64
    q = (rng.standard_normal((40, 2048)) * 1.1).astype(np.float32)
    scaled_ref = objs["scaler"].transform(q.copy())
    pca_ref = objs["pca"].transform(scaled_ref)
    scaled, pca = oracle.scaler_pca(params, q, acc64=True)
    assert np.array_equal(scaled, scaled_ref)
    H.assert_close_scaled(pca, pca_ref, H.TOL_STAGE, "pca")
    for key, p in (("Conservative", params.conservative), ("Moderate", params.moderate)):
        d = objs["detectors"][key]
        dec, pred = oracle.ocsvm_decision(p, pca_ref)
        tol = H.TOL_DEC_STAGE * np.abs(p.dual_coef).sum()
        assert np.abs(dec - d.decision_function(pca_ref)).max() <= tol
        H.flags_agree(dec, pred, d.decision_function(pca_ref), d.predict(pca_ref), tol, key)
        assert abs(np.abs(p.dual_coef).sum() - d.nu * len(feats)) < 1e-6 * len(feats)   # sum(alpha) = nu * N


def test_oracle_screen_uses_second_encoder():
    """improved_detection.py:125 vs :130: features come from encoder.keras, errors from the autoencoder."""
    g_ae = synth.random_cae(seed=11)
    enc = synth.perturbed_encoder(g_ae)
    x = synth.synth_crops(1, 0, 3)
    fa = oracle.cae_forward(g_ae, x, want=("features",))["features"]
    fe = oracle.cae_forward(enc, x, want=("features",))["features"]
    assert fe.shape == fa.shape == (3, 2048)
    assert np.abs(fa - fe).max() > 1e-4
