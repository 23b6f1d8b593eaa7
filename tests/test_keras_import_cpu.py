"""The pure-Python HDF5 reader and the `.keras` weight importer against fixtures written by the real HDF5 library
(tests/golden/make_golden_keras.py under the conda interpreter: h5py 3.3.0 / HDF5 1.10.6) in the Keras 3 saving
layout.  No Keras exists in this image: the layout is the published one, the bytes are real HDF5."""
import os
import pickle
import zipfile

import numpy as np
import pytest

from conftest import GOLDEN
from cellscreen import h5lite, model_io, synth


@pytest.fixture(scope="module")
def arrays():
    return np.load(os.path.join(GOLDEN, "golden_keras_like.npz"))


def test_old_style_groups_every_dataset(arrays):
    with zipfile.ZipFile(os.path.join(GOLDEN, "golden_keras_like.keras")) as z:
        tree = h5lite.read(z.read("model.weights.h5"))
    sfx = lambda base, k: base if k == 0 else f"{base}_{k}"
    for l in range(7):
        assert np.array_equal(tree[f"layers/{sfx('conv2d', l)}/vars/0"], arrays[f"conv{l}_kernel"])
        assert np.array_equal(tree[f"layers/{sfx('conv2d', l)}/vars/1"], arrays[f"conv{l}_bias"])
        if l < 6:
            for i, n in enumerate(("gamma", "beta", "mean", "var")):
                assert np.array_equal(tree[f"layers/{sfx('batch_normalization', l)}/vars/{i}"], arrays[f"bn{l}_{n}"])
    assert tree["optimizer/vars/0"] == 123 and tree["optimizer/vars/0"].dtype == np.int64
    assert len(tree) == 7 * 2 + 6 * 4 + 2


def test_new_style_groups_and_scalars(arrays):
    t = h5lite.read(os.path.join(GOLDEN, "golden_h5_v2_small.h5"))
    assert np.array_equal(t["layers/conv2d/vars/0"], arrays["conv1_kernel"])
    assert np.array_equal(t["layers/batch_normalization/vars/0"], arrays["bn1_gamma"])
    assert t["scalar"].shape == () and float(t["scalar"]) == 2.5
    assert np.array_equal(t["ints"], np.arange(6, dtype=np.int32).reshape(2, 3))


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 100)
    with pytest.raises(h5lite.H5Error):
        h5lite.read(str(p))


def test_cae_from_keras(arrays):
    w = model_io.cae_from_keras(os.path.join(GOLDEN, "golden_keras_like.keras"))
    assert w.n_conv == 7 and w.n_enc == 3 and w.input_hw == (64, 64) and abs(w.bn_eps - 1e-3) < 1e-12
    assert [k.shape[3] for k in w.kernels] == [4, 8, 4, 4, 8, 4, 1]
    for l in range(7):
        assert np.array_equal(w.kernels[l], arrays[f"conv{l}_kernel"]) and np.array_equal(w.biases[l], arrays[f"conv{l}_bias"])
    for l in range(6):
        assert np.array_equal(w.bn_gamma[l], arrays[f"bn{l}_gamma"]) and np.array_equal(w.bn_var[l], arrays[f"bn{l}_var"])
        assert np.array_equal(w.bn_beta[l], arrays[f"bn{l}_beta"]) and np.array_equal(w.bn_mean[l], arrays[f"bn{l}_mean"])


def test_encoder_archive(arrays):
    e = model_io.cae_from_keras(os.path.join(GOLDEN, "golden_keras_like_encoder.keras"))
    assert e.n_conv == 3 and e.n_enc == 3 and len(e.bn_gamma) == 3
    assert np.array_equal(e.kernels[2], arrays["conv2_kernel"]) and np.array_equal(e.bn_mean[2], arrays["bn2_mean"])


def test_convert_reference_model_dir(tmp_path, golden_det):
    """best_autoencoder.keras + encoder.keras + the four pickles -> native model dir, then back through the
    native reader: what load_trained_models (improved_detection.py:23-46) needs, with no Keras anywhere."""
    import shutil
    import helpers as H
    d = tmp_path / "models"
    d.mkdir()
    src = os.path.join(GOLDEN, "golden_keras_like.keras")
    shutil.copy(src, d / "best_autoencoder.keras")
    shutil.copy(os.path.join(GOLDEN, "golden_keras_like_encoder.keras"), d / "encoder.keras")
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    from sklearn.svm import OneClassSVM
    rng = np.random.default_rng(0)
    f = rng.standard_normal((200, 8 * 8 * 4)).astype(np.float32)            # the fixture's feature size
    sc = RobustScaler().fit(f)
    pc = PCA(n_components=20, random_state=0).fit(sc.transform(f))
    z = pc.transform(sc.transform(f))
    dets = [OneClassSVM(kernel="rbf", gamma="scale", nu=nu).fit(z) for nu in (0.05, 0.10)]
    for name, obj in (("scaler.pkl", sc), ("pca.pkl", pc), ("detector_conservative.pkl", dets[0]), ("detector_moderate.pkl", dets[1])):
        with open(d / name, "wb") as fh:
            pickle.dump(obj, fh)
    out = model_io.convert_reference_model_dir(str(d))
    ae, enc, det = model_io.load_model_dir(out)
    assert ae.n_conv == 7 and enc.n_conv == 3 and det.n_components == 20 and det.n_features == 256
    assert np.array_equal(ae.kernels[3], model_io.cae_from_keras(src).kernels[3])
    assert os.path.exists(os.path.join(out, "cae.bin")) and os.path.exists(os.path.join(out, "detector.bin"))
