"""Writers of the reference's model files (CAE_improved_modeltrain.py:271,299-300): the pure-Python HDF5 writer and the
`.keras` archive writer, round-tripped through the pure-Python reader bit for bit and opened with the REAL HDF5 library
(h5py of the image's conda interpreter, in a child process -- this interpreter has no h5py)."""
import json
import os
import pickle
import shutil
import subprocess
import zipfile

import numpy as np
import pytest

from cellscreen import h5lite, model_io, spec, synth

CONDA_PY = "/opt/conda/bin/python3.9"


def _same(a, b):
    assert a.n_conv == b.n_conv and a.n_enc == b.n_enc and tuple(a.input_hw) == tuple(b.input_hw) and abs(a.bn_eps - b.bn_eps) < 1e-9   # the native archive stores eps as float32
    for l in range(a.n_conv):
        assert np.array_equal(a.kernels[l], b.kernels[l]) and np.array_equal(a.biases[l], b.biases[l])
    for l in range(len(a.bn_gamma)):
        for f in ("bn_gamma", "bn_beta", "bn_mean", "bn_var"):
            assert np.array_equal(getattr(a, f)[l], getattr(b, f)[l]), (f, l)


def test_h5_writer_round_trip_every_dtype_and_shape():
    rng = np.random.default_rng(0)
    tree = {"a/b/c": rng.standard_normal((3, 3, 2, 5)).astype(np.float32), "a/b/d": rng.standard_normal(7),
            "a/s": np.float32(2.5), "i8": np.arange(12, dtype=np.int64).reshape(3, 4), "u1": np.arange(5, dtype=np.uint8),
            "empty/vars": {}, "z": np.zeros((0,), np.float32)}
    for k in range(20):                                       # more links than one symbol-table node holds
        tree[f"many/k{k:02d}"] = np.float32(k)
    back = h5lite.read(h5lite.write(tree))
    flat = {k: v for k, v in tree.items() if not isinstance(v, dict)}
    assert set(back) == set(flat)
    for k, v in flat.items():
        v = np.asarray(v)
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k


def test_keras_archives_round_trip_bit_identically(tmp_path):
    w = synth.random_cae(seed=3)
    model_io.cae_to_keras(str(tmp_path / "best_autoencoder.keras"), w)
    model_io.cae_to_keras(str(tmp_path / "encoder.keras"), w.encoder_half())
    _same(w, model_io.cae_from_keras(str(tmp_path / "best_autoencoder.keras")))
    _same(w.encoder_half(), model_io.cae_from_keras(str(tmp_path / "encoder.keras")))
    with zipfile.ZipFile(tmp_path / "best_autoencoder.keras") as z:
        assert set(z.namelist()) == {"metadata.json", "config.json", "model.weights.h5"}
        cfg = json.loads(z.read("config.json"))
    names = [l["config"]["name"] for l in cfg["config"]["layers"]]
    assert names[:5] == ["input_layer", "conv2d", "batch_normalization", "max_pooling2d", "conv2d_1"] and len(names) == 20
    assert cfg["config"]["layers"][-1]["config"]["activation"] == "sigmoid" and cfg["class_name"] == "Functional"


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py in this image")
def test_written_weights_open_with_the_real_hdf5_library(tmp_path):
    w = synth.random_cae(seed=4)
    path = str(tmp_path / "final_autoencoder.keras")
    model_io.cae_to_keras(path, w)
    np.savez(tmp_path / "want.npz", **{f"conv{l}_kernel": w.kernels[l] for l in range(7)}, **{f"bn{l}_var": w.bn_var[l] for l in range(6)})
    code = r"""
import io, sys, zipfile, h5py, numpy as np
z = zipfile.ZipFile(sys.argv[1]); want = np.load(sys.argv[2])
f = h5py.File(io.BytesIO(z.read("model.weights.h5")), "r")
sfx = lambda b, k: b if k == 0 else f"{b}_{k}"
for l in range(7):
    d = f[f"layers/{sfx('conv2d', l)}/vars/0"]
    assert d.dtype == np.float32 and np.array_equal(d[()], want[f"conv{l}_kernel"]), l
for l in range(6):
    assert np.array_equal(f[f"layers/{sfx('batch_normalization', l)}/vars/3"][()], want[f"bn{l}_var"]), l
assert len(f["layers"]) == 20 and len(f["layers/max_pooling2d/vars"]) == 0
print("ok", h5py.version.hdf5_version)
"""
    r = subprocess.run([CONDA_PY, "-c", code, path, str(tmp_path / "want.npz")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_six_reference_files_are_converted_in_place_on_load(tmp_path, golden_det):
    """ensure_native_model_dir: a directory holding only the reference's six files (improved_detection.py:28-41) gets the
    native set written beside them; a native set older than the reference files is refreshed."""
    import helpers as H
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    from sklearn.svm import OneClassSVM
    w = synth.random_cae(seed=5)
    d = tmp_path / "models"
    d.mkdir()
    model_io.cae_to_keras(str(d / "best_autoencoder.keras"), w)
    model_io.cae_to_keras(str(d / "encoder.keras"), synth.perturbed_encoder(w))
    rng = np.random.default_rng(0)
    f = rng.standard_normal((150, 2048)).astype(np.float32)
    sc = RobustScaler().fit(f)
    pc = PCA(n_components=10, random_state=0).fit(sc.transform(f))
    zed = pc.transform(sc.transform(f))
    dets = [OneClassSVM(kernel="rbf", gamma="scale", nu=nu).fit(zed) for nu in (0.05, 0.10)]
    for name, obj in (("scaler.pkl", sc), ("pca.pkl", pc), ("detector_conservative.pkl", dets[0]), ("detector_moderate.pkl", dets[1])):
        with open(d / name, "wb") as fh:
            pickle.dump(obj, fh)
    assert model_io.has_reference_files(str(d)) and not model_io.has_native_files(str(d))
    assert model_io.ensure_native_model_dir(str(d)) == str(d) and model_io.has_native_files(str(d))
    ae, enc, det = model_io.load_model_dir(str(d))
    _same(w, ae)
    assert enc is not None and det.n_components == 10 and not np.array_equal(enc.kernels[0], ae.kernels[0])
    t0 = os.path.getmtime(d / spec.NATIVE_CAE)
    assert model_io.ensure_native_model_dir(str(d)) == str(d) and os.path.getmtime(d / spec.NATIVE_CAE) == t0    # up to date: untouched
    w2 = synth.random_cae(seed=6)
    model_io.cae_to_keras(str(d / "best_autoencoder.keras"), w2)
    os.utime(d / "best_autoencoder.keras", (t0 + 10, t0 + 10))
    model_io.ensure_native_model_dir(str(d))
    _same(w2, model_io.load_model_dir(str(d))[0])
    only_native = tmp_path / "native"
    shutil.copytree(d, only_native)
    for f_ in spec.REF_MODEL_FILES:
        os.remove(only_native / f_)
    assert model_io.ensure_native_model_dir(str(only_native)) == str(only_native)
