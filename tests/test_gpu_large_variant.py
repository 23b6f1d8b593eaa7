"""BASELINE.json configs[4] as a parity case: the reference's layer grammar at another size --
128x128 crops, filters 32-64-128 | 128-64-32-1 (create_improved_autoencoder
takes input_shape, CAE_improved_modeltrain.py:184) -- through the same C ABI, on the run-time-shaped
MFMA kernels of csrc/conv_generic.hip, against the CPU oracle (fp64 accumulation) at the tolerances
of the reference architecture (helpers.py).  Plus a rectangular variant and the refusal paths."""
import numpy as np
import pytest

import helpers as H
from cellscreen import _lib as L
from cellscreen import spec, synth
from cellscreen.detector_fit import fit_detector
from cellscreen.engine import Engine
from oracle import oracle

pytestmark = pytest.mark.gpu

LARGE_HW = (128, 128)
LARGE_CH = (32, 64, 128, 128, 64, 32, 1)


@pytest.fixture(scope="module")
def large():
    return synth.random_cae(seed=5, hw=LARGE_HW, channels=LARGE_CH, n_enc=3)


@pytest.fixture(scope="module")
def crops():
    return np.concatenate([synth.synth_crops(42, 0, 6, hw=LARGE_HW), synth.blob_crops(4, 6, hw=LARGE_HW)])


def test_large_variant_layers_and_reconstruction(large, crops):
    e = Engine.from_weights(large)
    try:
        # automatic pass size for device-resident input: what a ~28 GB workspace holds (1.9 MB of activations per cell here)
        assert e.info.chunk_cells == 14336
        assert (e.info.height, e.info.width, e.info.n_conv, e.info.n_enc) == (128, 128, 7, 3)
        assert e.info.reference_arch == 0 and e.info.feature_dim == 16 * 16 * 128
        assert list(e.info.channels[:7]) == list(LARGE_CH)
        ref = oracle.cae_forward(large, crops, acc64=True, layers=True)
        for l in range(7):
            got = e.layer_output(crops, l)
            want = ref["layers"][l].reshape(got.shape)
            H.assert_close_scaled(got, want, H.TOL_FEATURES if l < 6 else H.TOL_RECON, f"layer {l}")
        rec, mse, mae = e.reconstruct(crops)
        assert np.abs(rec - ref["recon"].reshape(rec.shape)).max() <= H.TOL_RECON
        H.assert_rel(mse, ref["mse"], H.TOL_ERR_REL, "mse")
        H.assert_rel(mae, ref["mae"], H.TOL_ERR_REL, "mae")
        f = e.encode(crops)
        assert f.shape == (len(crops), 32768)
        H.assert_close_scaled(f, ref["features"].reshape(f.shape), H.TOL_FEATURES, "features (h,w,c)")
    finally:
        e.close()


def test_large_variant_screen_end_to_end(large):
    """Detector fitted (real scikit-learn) on GPU-encoded features of the large model; scores and flags
    against the oracle pipeline."""
    train = synth.blob_crops(11, 160, hw=LARGE_HW)
    e0 = Engine.from_weights(large)
    feats = e0.encode(train)
    e0.close()
    det, _ = fit_detector(feats, pca_random_state=0)
    assert det.n_features == 32768
    x = np.concatenate([synth.blob_crops(12, 20, hw=LARGE_HW), synth.synth_crops(7, 0, 4, hw=LARGE_HW)])
    e = Engine.from_weights(large, None, det)
    try:
        r = e.screen(x)
        ref = oracle.screen(large, None, det, x, acc64=True)
        H.assert_rel(r["mse"], ref["mse"], H.TOL_ERR_REL, "mse")
        for name, p in (("cons", det.conservative), ("mod", det.moderate)):
            tol = H.TOL_DEC_E2E * np.abs(p.dual_coef).sum()
            assert np.abs(r[f"{name}_score"] - ref[f"{name}_score"]).max() <= tol, name
            H.flags_agree(-r[f"{name}_score"], r[f"{name}_pred"], ref[f"{name}_dec"], ref[f"{name}_pred"], tol, name)
        # chunking and device-resident input give the same bits
        import torch
        e.set_chunk(7)
        r2 = e.screen(torch.from_numpy(x).cuda())
        assert np.array_equal(r2["mse"].cpu().numpy(), r["mse"]) and np.array_equal(r2["mod_score"].cpu().numpy(), r["mod_score"])
    finally:
        e.close()


def test_rectangular_variant_with_odd_channel_counts():
    w = synth.random_cae(seed=9, hw=(64, 128), channels=(16, 32, 48, 48, 32, 16, 1), n_enc=3)
    x = synth.synth_crops(3, 0, 5, hw=(64, 128))
    e = Engine.from_weights(w)
    try:
        ref = oracle.cae_forward(w, x, acc64=True)
        rec, mse, mae = e.reconstruct(x)
        assert np.abs(rec - ref["recon"].reshape(rec.shape)).max() <= H.TOL_RECON
        H.assert_rel(mse, ref["mse"], H.TOL_ERR_REL, "mse")
        H.assert_close_scaled(e.encode(x), ref["features"].reshape(5, -1), H.TOL_FEATURES, "features")
    finally:
        e.close()


def test_separate_encoder_on_the_generic_path(large, crops):
    enc = synth.perturbed_encoder(large)
    e = Engine.from_weights(large, enc)
    try:
        assert e.info.shared_encoder == 0
        f1 = e.encode(crops[:4], which=1)
        ref = oracle.cae_forward(enc, crops[:4], acc64=True, want=("features",))
        H.assert_close_scaled(f1, ref["features"].reshape(4, -1), H.TOL_FEATURES, "encoder.keras features")
        f0 = e.encode(crops[:4], which=0)
        assert not np.array_equal(f0, f1)
    finally:
        e.close()


def test_unsupported_shapes_are_refused():
    with pytest.raises(L.CellScreenError) as ei:                 # conv4 grid 8 wide: below the generic kernel's 16
        Engine.from_weights(synth.random_cae(seed=1, hw=(64, 64), channels=(16, 32, 64, 64, 32, 16, 1), n_enc=3))
    assert ei.value.status == -6
    with pytest.raises(L.CellScreenError) as ei:                 # not the reference grammar
        Engine.from_weights(synth.random_cae(seed=1, hw=(64, 64), channels=(32, 64, 32, 32, 1), n_enc=3))
    assert ei.value.status == -6
    from cellscreen.trainer import Trainer
    with pytest.raises(L.CellScreenError):                       # training exists for the reference graph only
        Trainer(synth.random_cae(seed=5, hw=LARGE_HW, channels=LARGE_CH, n_enc=3))
