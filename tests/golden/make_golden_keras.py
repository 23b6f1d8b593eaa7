"""Generates the `.keras`-shaped fixtures for the pure-Python HDF5 reader (cellscreen/h5lite.py) with a REAL
HDF5 library:   /opt/conda/bin/python3.9 tests/golden/make_golden_keras.py      (h5py 3.3.0 / HDF5 1.10.6)

Keras itself is not installed anywhere in this image, so the files follow the Keras 3 saving layout as
published (keras/src/saving/saving_lib.py): a zip with metadata.json, config.json and model.weights.h5, the
variables of layer k stored as datasets "<container>/<snake_case class name>[_<n>]/vars/<i>" by plain
`group[name] = array` assignments.  Small channel counts keep the fixture tiny; the graph is the reference's
(conv+BN+pool x3, conv+BN, (up+conv+BN) x2, up+conv) -- CAE_improved_modeltrain.py:188-216.
Writes golden_keras_like.keras (default h5py settings: old-style groups), golden_h5_v2_small.h5
(libver='latest': version-2 object headers, compact link messages) and golden_keras_like.npz (the arrays)."""
import io
import json
import os
import zipfile

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CH = (4, 8, 4, 4, 8, 4, 1)
N_ENC = 3


def arrays():
    rng = np.random.default_rng(7)
    out, cin = {}, 1
    for l, c in enumerate(CH):
        out[f"conv{l}_kernel"] = rng.standard_normal((3, 3, cin, c)).astype(np.float32)
        out[f"conv{l}_bias"] = rng.standard_normal(c).astype(np.float32)
        if l < len(CH) - 1:
            out[f"bn{l}_gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            out[f"bn{l}_beta"] = rng.standard_normal(c).astype(np.float32)
            out[f"bn{l}_mean"] = rng.standard_normal(c).astype(np.float32)
            out[f"bn{l}_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        cin = c
    return out


def write_weights(f, a, n_layers=len(CH)):
    sfx = lambda base, k: base if k == 0 else f"{base}_{k}"
    for l in range(n_layers):
        g = f.create_group(f"layers/{sfx('conv2d', l)}").create_group("vars")
        g["0"] = a[f"conv{l}_kernel"]
        g["1"] = a[f"conv{l}_bias"]
        if l < len(CH) - 1:
            g = f.create_group(f"layers/{sfx('batch_normalization', l)}").create_group("vars")
            for i, n in enumerate(("gamma", "beta", "mean", "var")):
                g[str(i)] = a[f"bn{l}_{n}"]
    for k in range(3):                                   # layers without variables still get their (empty) groups
        f.create_group(f"layers/{sfx('max_pooling2d', k)}/vars")
        f.create_group(f"layers/{sfx('up_sampling2d', k)}/vars")
    f.create_group("layers/input_layer/vars")
    o = f.create_group("optimizer/vars")                 # Adam slots: must be ignored by the importer
    o["0"] = np.int64(123)
    o["1"] = np.float32(1e-3)


def config():
    layers = [{"class_name": "InputLayer", "config": {"batch_shape": [None, 64, 64, 1], "name": "input_layer"}}]
    k = 0
    for l in range(len(CH)):
        if l > N_ENC:
            layers.append({"class_name": "UpSampling2D", "config": {"name": f"up_sampling2d_{l - N_ENC - 1}" if l > N_ENC + 1 else "up_sampling2d"}})
        layers.append({"class_name": "Conv2D", "config": {"name": "conv2d" if l == 0 else f"conv2d_{l}", "filters": CH[l]}})
        if l < len(CH) - 1:
            layers.append({"class_name": "BatchNormalization", "config": {"name": "batch_normalization" if l == 0 else f"batch_normalization_{l}",
                                                                         "epsilon": 0.001, "momentum": 0.99}})
        if l < N_ENC:
            layers.append({"class_name": "MaxPooling2D", "config": {"name": "max_pooling2d" if l == 0 else f"max_pooling2d_{l}"}})
    return {"class_name": "Functional", "config": {"name": "functional", "layers": layers}}


def main():
    a = arrays()
    bio = io.BytesIO()
    with h5py.File(bio, "w") as f:
        write_weights(f, a)
    with zipfile.ZipFile(os.path.join(HERE, "golden_keras_like.keras"), "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "3.x (layout only; written by h5py)", "date_saved": "fixture"}))
        z.writestr("config.json", json.dumps(config()))
        z.writestr("model.weights.h5", bio.getvalue())
    # encoder.keras: Model(input, encoded) -- the first N_ENC conv + BN (+ pool) layers (CAE_improved_modeltrain.py:220)
    bio = io.BytesIO()
    with h5py.File(bio, "w") as f:
        write_weights(f, a, N_ENC)
    cfg = config()
    cfg["config"]["layers"] = [l for l in cfg["config"]["layers"] if l["class_name"] != "UpSampling2D"][:1 + 3 * N_ENC]
    with zipfile.ZipFile(os.path.join(HERE, "golden_keras_like_encoder.keras"), "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "3.x (layout only; written by h5py)", "date_saved": "fixture"}))
        z.writestr("config.json", json.dumps(cfg))
        z.writestr("model.weights.h5", bio.getvalue())
    # new-style groups (version-2 object headers, compact link messages): what libver='latest' writes for groups
    # of at most 8 links.  Keras opens its file with h5py's defaults (old-style groups, as above); larger new-style
    # groups use dense link storage, which the reader refuses with a clear error.
    with h5py.File(os.path.join(HERE, "golden_h5_v2_small.h5"), "w", libver="latest") as f:
        g = f.create_group("layers/conv2d/vars")
        g["0"] = a["conv1_kernel"]
        g["1"] = a["conv1_bias"]
        f.create_group("layers/batch_normalization/vars")["0"] = a["bn1_gamma"]
        f["scalar"] = np.float64(2.5)
        f["ints"] = np.arange(6, dtype=np.int32).reshape(2, 3)
    np.savez_compressed(os.path.join(HERE, "golden_keras_like.npz"), **a)
    for n in ("golden_keras_like.keras", "golden_keras_like_encoder.keras", "golden_h5_v2_small.h5", "golden_keras_like.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)))


if __name__ == "__main__":
    main()
