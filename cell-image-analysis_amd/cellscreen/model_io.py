"""model_dir persistence -- the load side of ProductionMutantScreening.load_trained_models
(improved_detection.py:23-46) and the save side of the trainer
(CAE_improved_modeltrain.py:270-275, 299-300, 437-444).

Native format: <model_dir>/cae.bin + detector.bin + manifest.json.  The .bin files are
"tensor archives" (layout in csrc/tensor_archive.hpp) that libcellscreen reads itself
(cs_model_load).  The reference's own files can be converted when their libraries are
importable: sklearn pickles via pickle (sklearn is installed), Keras weights from a
.npz export (h5py/Keras are not installed here, so .keras itself is not readable).
"""
from __future__ import annotations

import json
import os
import pickle
import struct
from typing import Dict, Optional

import numpy as np

from . import spec
from .spec import CAEWeights, DetectorParams, OCSVMParams

_MAGIC = b"CSTENS01"
_DTYPES = {np.dtype("float32"): 0, np.dtype("float64"): 1, np.dtype("int32"): 2, np.dtype("int64"): 3}
_RDTYPES = {v: k for k, v in _DTYPES.items()}


def write_archive(path: str, tensors: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(_MAGIC)
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr)
            if a.dtype not in _DTYPES:
                raise TypeError(f"{name}: unsupported dtype {a.dtype}")
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<II", _DTYPES[a.dtype], a.ndim))
            for d in a.shape:
                f.write(struct.pack("<Q", d))
            f.write(struct.pack("<Q", a.nbytes))
            f.write(b"\0" * ((8 - f.tell() % 8) % 8))
            f.write(a.tobytes())
            f.write(b"\0" * ((8 - f.tell() % 8) % 8))


def read_archive(path: str) -> Dict[str, np.ndarray]:
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != _MAGIC:
            raise ValueError(f"{path}: bad magic")
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            dt, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<" + "Q" * nd, f.read(8 * nd)) if nd else ()
            (nbytes,) = struct.unpack("<Q", f.read(8))
            f.seek((8 - f.tell() % 8) % 8, 1)
            data = f.read(nbytes)
            if len(data) != nbytes:
                raise ValueError(f"{path}: truncated tensor {name}")
            f.seek((8 - f.tell() % 8) % 8, 1)
            out[name] = np.frombuffer(data, dtype=_RDTYPES[dt]).reshape(dims).copy()
    return out


def _cae_tensors(prefix: str, w: CAEWeights) -> Dict[str, np.ndarray]:
    t = {}
    for l in range(w.n_conv):
        t[f"{prefix}.conv{l}.kernel"] = w.kernels[l].astype(np.float32)
        t[f"{prefix}.conv{l}.bias"] = w.biases[l].astype(np.float32)
    for l in range(len(w.bn_gamma)):
        t[f"{prefix}.bn{l}.gamma"] = w.bn_gamma[l].astype(np.float32)
        t[f"{prefix}.bn{l}.beta"] = w.bn_beta[l].astype(np.float32)
        t[f"{prefix}.bn{l}.mean"] = w.bn_mean[l].astype(np.float32)
        t[f"{prefix}.bn{l}.var"] = w.bn_var[l].astype(np.float32)
    return t


def _cae_from_tensors(prefix: str, t: Dict[str, np.ndarray], n_conv: int, n_bn: int, hw, n_enc, eps) -> CAEWeights:
    return CAEWeights(
        kernels=[t[f"{prefix}.conv{l}.kernel"] for l in range(n_conv)],
        biases=[t[f"{prefix}.conv{l}.bias"] for l in range(n_conv)],
        bn_gamma=[t[f"{prefix}.bn{l}.gamma"] for l in range(n_bn)],
        bn_beta=[t[f"{prefix}.bn{l}.beta"] for l in range(n_bn)],
        bn_mean=[t[f"{prefix}.bn{l}.mean"] for l in range(n_bn)],
        bn_var=[t[f"{prefix}.bn{l}.var"] for l in range(n_bn)],
        input_hw=tuple(hw), n_enc=n_enc, bn_eps=eps).validate()


def save_model_dir(model_dir: str, autoencoder: CAEWeights, encoder: Optional[CAEWeights] = None,
                   detector: Optional[DetectorParams] = None, extra: Optional[dict] = None) -> None:
    """Writes the native file set.  `encoder` is the encoder.keras weight set
    (CAE_improved_modeltrain.py:300); omit it when it equals the autoencoder's encoder half."""
    os.makedirs(model_dir, exist_ok=True)
    autoencoder.validate()
    t = {"meta": np.array([autoencoder.input_hw[0], autoencoder.input_hw[1], autoencoder.n_conv,
                           autoencoder.n_enc, *autoencoder.channels], dtype=np.int32),
         "bn_eps": np.array([autoencoder.bn_eps], dtype=np.float32)}
    t.update(_cae_tensors("ae", autoencoder))
    if encoder is not None:
        t.update(_cae_tensors("enc", encoder))
    write_archive(os.path.join(model_dir, spec.NATIVE_CAE), t)
    if detector is not None:
        d = {"scaler.center": detector.scaler_center.astype(np.float32),
             "scaler.scale": detector.scaler_scale.astype(np.float64),
             "pca.components": detector.pca_components.astype(np.float32),
             "pca.mean": detector.pca_mean.astype(np.float32),
             "pca.mean_proj": detector.pca_mean_proj.astype(np.float32)}
        for name, p in (("svm_conservative", detector.conservative), ("svm_moderate", detector.moderate)):
            d[f"{name}.sv"] = np.ascontiguousarray(p.support_vectors, dtype=np.float64)
            d[f"{name}.dual_coef"] = np.ascontiguousarray(p.dual_coef, dtype=np.float64).ravel()
            d[f"{name}.gamma"] = np.array([p.gamma], dtype=np.float64)
            d[f"{name}.rho"] = np.array([p.rho], dtype=np.float64)
        write_archive(os.path.join(model_dir, spec.NATIVE_DETECTOR), d)
    manifest = {"format": "cellscreen-model-dir", "version": 1,
                "input_hw": list(autoencoder.input_hw), "channels": list(autoencoder.channels),
                "n_enc": autoencoder.n_enc, "separate_encoder": encoder is not None,
                "has_detector": detector is not None}
    if detector is not None:
        manifest.update(n_components=detector.n_components, n_sv_conservative=detector.conservative.n_sv,
                        n_sv_moderate=detector.moderate.n_sv)
    if extra:
        manifest.update(extra)
    with open(os.path.join(model_dir, spec.NATIVE_MANIFEST), "w") as f:
        json.dump(manifest, f, indent=1)


def load_model_dir(model_dir: str):
    """-> (autoencoder, encoder_or_None, detector_or_None) as numpy containers."""
    t = read_archive(os.path.join(model_dir, spec.NATIVE_CAE))
    meta = t["meta"]
    hw, n_conv, n_enc = (int(meta[0]), int(meta[1])), int(meta[2]), int(meta[3])
    eps = float(t["bn_eps"][0])
    ae = _cae_from_tensors("ae", t, n_conv, n_conv - 1, hw, n_enc, eps)
    enc = _cae_from_tensors("enc", t, n_enc, n_enc, hw, n_enc, eps) if "enc.conv0.kernel" in t else None
    det = None
    dpath = os.path.join(model_dir, spec.NATIVE_DETECTOR)
    if os.path.exists(dpath):
        d = read_archive(dpath)
        svm = lambda n: OCSVMParams(d[f"{n}.sv"], d[f"{n}.dual_coef"], float(d[f"{n}.gamma"][0]), float(d[f"{n}.rho"][0]))
        det = DetectorParams(d["scaler.center"], d["scaler.scale"], d["pca.components"], d["pca.mean"],
                             d["pca.mean_proj"], svm("svm_conservative"), svm("svm_moderate"))
    return ae, enc, det


# ---- conversion from the reference's sklearn objects (CAE_improved_modeltrain.py:437-444) ----
def ocsvm_params_from_sklearn(det) -> OCSVMParams:
    """sklearn.svm.OneClassSVM -> arrays.  decision_function = sum_i dual_coef_i k(x, sv_i) +
    intercept_ (sklearn svm/_base.py), libsvm's rho = -intercept_ = offset_ (_classes.py:1735)."""
    if det.kernel != "rbf":
        raise ValueError("only kernel='rbf' is on the reference path (CAE_improved_modeltrain.py:421)")
    return OCSVMParams(np.ascontiguousarray(det.support_vectors_, dtype=np.float64),
                       np.ascontiguousarray(det.dual_coef_, dtype=np.float64).ravel(),
                       float(det._gamma), float(-det.intercept_[0]))


def detector_params_from_sklearn(scaler, pca, det_conservative, det_moderate) -> DetectorParams:
    comps = np.ascontiguousarray(pca.components_)
    mean = np.asarray(pca.mean_)
    if pca.whiten:
        raise ValueError("PCA(whiten=True) is not on the reference path")
    # the exact expression PCA.transform evaluates on every call (sklearn _base.py:147-155)
    mean_proj = (mean.reshape(1, -1) @ comps.T).ravel()
    return DetectorParams(np.asarray(scaler.center_), np.asarray(scaler.scale_, dtype=np.float64),
                          comps, mean, mean_proj,
                          ocsvm_params_from_sklearn(det_conservative), ocsvm_params_from_sklearn(det_moderate))


def detector_from_reference_pickles(model_dir: str) -> DetectorParams:
    """Reads scaler.pkl / pca.pkl / detector_*.pkl as improved_detection.py:32-41 does."""
    def ld(name):
        with open(os.path.join(model_dir, name), "rb") as f:
            return pickle.load(f)
    return detector_params_from_sklearn(ld("scaler.pkl"), ld("pca.pkl"), ld("detector_conservative.pkl"),
                                        ld("detector_moderate.pkl"))


def cae_from_npz(path: str, prefix: str = "") -> CAEWeights:
    """Keras weights exported as arrays named conv{l}_kernel, conv{l}_bias, bn{l}_gamma, ..."""
    z = np.load(path)
    n_conv = sum(1 for k in z.files if k.startswith(prefix + "conv") and k.endswith("_kernel"))
    n_bn = sum(1 for k in z.files if k.startswith(prefix + "bn") and k.endswith("_gamma"))
    g = lambda n: np.asarray(z[prefix + n], dtype=np.float32)
    return CAEWeights([g(f"conv{l}_kernel") for l in range(n_conv)], [g(f"conv{l}_bias") for l in range(n_conv)],
                      [g(f"bn{l}_gamma") for l in range(n_bn)], [g(f"bn{l}_beta") for l in range(n_bn)],
                      [g(f"bn{l}_mean") for l in range(n_bn)], [g(f"bn{l}_var") for l in range(n_bn)]).validate()


# ---- `.keras` archives (best_autoencoder.keras, encoder.keras of improved_detection.py:28-29) --------------
def _suffix_key(name: str):
    """Keras names the groups of a layer container <snake_case class>[_<n>] in container order:
    conv2d, conv2d_1, conv2d_2, ... -> sort key (0, 1, 2, ...)."""
    head, _, tail = name.rpartition("_")
    return int(tail) if head and tail.isdigit() else 0


def cae_from_keras(path: str) -> CAEWeights:
    """Reads a Keras-3 `.keras` archive (zip: config.json + model.weights.h5) -- or a bare weights .h5 -- of the
    reference's autoencoder or encoder WITHOUT Keras or h5py (cellscreen/h5lite.py parses the HDF5 file).
    Layers are identified by what they store, not by their path prefix: a group whose `vars` holds a 4-D "0" and a
    1-D "1" is a Conv2D (kernel HWIO, bias); one with four 1-D arrays is a BatchNormalization (gamma, beta,
    moving_mean, moving_variance); optimizer slots and everything else are ignored.  Order = the numeric suffix of
    the group name (conv2d, conv2d_1, ...), which is the model's layer order.  Input size, pooling count and the BN
    epsilon come from config.json when present (else 64x64, the layer grammar's n_enc, Keras's 1e-3).
    The layout follows the published Keras 3 saving code; no Keras install exists here to cross-check a real file --
    the fixture under tests/golden is written with the real HDF5 library in that layout."""
    import json
    import zipfile
    from . import h5lite
    cfg = None
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
            wname = next((n for n in names if n.endswith("model.weights.h5")), None)
            if wname is None:
                raise ValueError(f"{path}: no model.weights.h5 inside the archive (found {names})")
            tree = h5lite.read(z.read(wname))
            if "config.json" in names:
                cfg = json.loads(z.read("config.json"))
    else:
        tree = h5lite.read(path)
    layers: Dict[str, Dict[int, np.ndarray]] = {}
    for key, arr in tree.items():
        parts = key.split("/")
        if len(parts) >= 3 and parts[-2] == "vars" and parts[-1].isdigit() and "optimizer" not in parts[:-2]:
            layers.setdefault("/".join(parts[:-2]), {})[int(parts[-1])] = arr
    convs, bns = [], []
    for lpath, v in layers.items():
        name = lpath.split("/")[-1]
        if len(v) == 2 and v.get(0) is not None and v[0].ndim == 4 and v[1].ndim == 1:
            convs.append((_suffix_key(name), name, v))
        elif len(v) == 4 and all(v[i].ndim == 1 for i in range(4)):
            bns.append((_suffix_key(name), name, v))
    convs.sort(key=lambda t: t[0]); bns.sort(key=lambda t: t[0])
    if not convs or len(bns) not in (len(convs), len(convs) - 1):
        raise ValueError(f"{path}: found {len(convs)} Conv2D and {len(bns)} BatchNormalization variable groups; "
                         "not the reference's conv/BN chain")
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    hw, n_pool, eps = spec.INPUT_HW, None, spec.BN_EPS
    if cfg is not None:
        try:
            ls = cfg["config"]["layers"]
            n_pool = sum(1 for l in ls if l.get("class_name") == "MaxPooling2D")
            for l in ls:
                c = l.get("config", {})
                shp = c.get("batch_shape") or c.get("batch_input_shape")
                if l.get("class_name") == "InputLayer" and shp and len(shp) == 4:
                    hw = (int(shp[1]), int(shp[2]))
                if l.get("class_name") == "BatchNormalization" and "epsilon" in c:
                    eps = float(c["epsilon"])
        except (KeyError, TypeError):
            pass
    encoder_only = len(bns) == len(convs)
    n_enc = n_pool if n_pool else (len(convs) if encoder_only else (len(convs) - 1) // 2)
    return CAEWeights([f32(v[0]) for _, _, v in convs], [f32(v[1]) for _, _, v in convs],
                      [f32(v[0]) for _, _, v in bns], [f32(v[1]) for _, _, v in bns],
                      [f32(v[2]) for _, _, v in bns], [f32(v[3]) for _, _, v in bns], hw, n_enc, eps).validate()


def _keras_layer_names(n_conv: int, n_enc: int, encoder_only: bool):
    """Layer sequence of the reference graph (CAE_improved_modeltrain.py:188-216) with the names Keras 3 gives layers
    created in that order in a fresh session: [(class_name, name, conv/bn index or None)]."""
    cnt: Dict[str, int] = {}

    def nm(base):
        k = cnt.get(base, 0)
        cnt[base] = k + 1
        return base if k == 0 else f"{base}_{k}"
    seq = [("InputLayer", nm("input_layer"), None)]
    for l in range(n_conv):
        if l > n_enc:
            seq.append(("UpSampling2D", nm("up_sampling2d"), None))
        seq.append(("Conv2D", nm("conv2d"), l))
        if l < n_conv - 1 or encoder_only:
            seq.append(("BatchNormalization", nm("batch_normalization"), l))
        if l < n_enc:
            seq.append(("MaxPooling2D", nm("max_pooling2d"), None))
    return seq


def _keras_config(w: CAEWeights, seq) -> dict:
    """config.json of the Functional model in the Keras 3 serialization layout (keras/src/models/functional.py,
    saving/serialization_lib.py as published).  Not cross-checked against a Keras install (none exists here)."""
    H, W = w.input_hw
    layers = []
    prev = None
    for cls, name, idx in seq:
        if cls == "InputLayer":
            cfg = {"batch_shape": [None, H, W, 1], "dtype": "float32", "sparse": False, "name": name}
        elif cls == "Conv2D":
            last = idx == w.n_conv - 1 and len(w.bn_gamma) < w.n_conv
            cfg = {"name": name, "trainable": True, "dtype": "float32", "filters": int(w.kernels[idx].shape[3]), "kernel_size": [3, 3],
                   "strides": [1, 1], "padding": "same", "data_format": "channels_last", "dilation_rate": [1, 1], "groups": 1,
                   "activation": "sigmoid" if last else "relu", "use_bias": True,
                   "kernel_initializer": {"module": "keras.initializers", "class_name": "GlorotUniform", "config": {"seed": None}, "registered_name": None},
                   "bias_initializer": {"module": "keras.initializers", "class_name": "Zeros", "config": {}, "registered_name": None},
                   "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}
        elif cls == "BatchNormalization":
            cfg = {"name": name, "trainable": True, "dtype": "float32", "axis": -1, "momentum": spec.BN_MOMENTUM, "epsilon": float(w.bn_eps),
                   "center": True, "scale": True,
                   "beta_initializer": {"module": "keras.initializers", "class_name": "Zeros", "config": {}, "registered_name": None},
                   "gamma_initializer": {"module": "keras.initializers", "class_name": "Ones", "config": {}, "registered_name": None},
                   "moving_mean_initializer": {"module": "keras.initializers", "class_name": "Zeros", "config": {}, "registered_name": None},
                   "moving_variance_initializer": {"module": "keras.initializers", "class_name": "Ones", "config": {}, "registered_name": None},
                   "beta_regularizer": None, "gamma_regularizer": None, "beta_constraint": None, "gamma_constraint": None, "synchronized": False}
        elif cls == "MaxPooling2D":
            cfg = {"name": name, "trainable": True, "dtype": "float32", "pool_size": [2, 2], "padding": "same", "strides": [2, 2],
                   "data_format": "channels_last"}
        else:
            cfg = {"name": name, "trainable": True, "dtype": "float32", "size": [2, 2], "data_format": "channels_last", "interpolation": "nearest"}
        entry = {"module": "keras.layers", "class_name": cls, "config": cfg, "registered_name": None, "name": name,
                 "inbound_nodes": [] if prev is None else [{"args": [{"class_name": "__keras_tensor__", "config": {"keras_history": [prev, 0, 0]}}], "kwargs": {}}]}
        layers.append(entry)
        prev = name
    return {"module": "keras.src.models.functional", "class_name": "Functional",
            "config": {"name": "functional", "trainable": True, "layers": layers,
                       "input_layers": [[seq[0][1], 0, 0]], "output_layers": [[prev, 0, 0]]},
            "registered_name": "Functional",
            "build_config": {"input_shape": None},
            "compile_config": {"optimizer": {"module": "keras.optimizers", "class_name": "Adam",
                                             "config": {"name": "adam", "learning_rate": spec.ADAM_LR, "beta_1": spec.ADAM_B1, "beta_2": spec.ADAM_B2,
                                                        "epsilon": spec.ADAM_EPS, "amsgrad": False}, "registered_name": None},
                               "loss": "mse", "loss_weights": None, "metrics": ["mae"], "weighted_metrics": None, "run_eagerly": False,
                               "steps_per_execution": 1, "jit_compile": False}}


def cae_to_keras(path: str, w: CAEWeights) -> None:
    """Writes a weight set as a Keras-3 `.keras` archive -- the files the reference's trainer leaves
    (`best_autoencoder.keras`, `final_autoencoder.keras`: the full autoencoder; `encoder.keras`: a weight set with
    n_conv == n_enc; CAE_improved_modeltrain.py:271,299-300) -- without Keras or h5py: zip of metadata.json, config.json
    and model.weights.h5 (cellscreen/h5lite.py writes the HDF5 bytes; variables of layer k as
    layers/<snake_case class>[_<n>]/vars/<i>, Conv2D = [kernel HWIO, bias], BatchNormalization = [gamma, beta, moving_mean,
    moving_variance], empty `vars` groups for the weightless layers).  Reads back bit-identically through cae_from_keras
    and opens with the real HDF5 library (tests); the JSON follows the published Keras 3 serialization and has not been
    loaded by a Keras install (there is none here)."""
    import zipfile
    from datetime import datetime
    from . import h5lite
    w.validate()
    encoder_only = w.n_conv == w.n_enc
    seq = _keras_layer_names(w.n_conv, w.n_enc, encoder_only)
    tree: Dict[str, object] = {}
    for cls, name, idx in seq:
        if cls == "Conv2D":
            tree[f"layers/{name}/vars/0"] = np.ascontiguousarray(w.kernels[idx], dtype=np.float32)
            tree[f"layers/{name}/vars/1"] = np.ascontiguousarray(w.biases[idx], dtype=np.float32)
        elif cls == "BatchNormalization":
            for i, a in enumerate((w.bn_gamma[idx], w.bn_beta[idx], w.bn_mean[idx], w.bn_var[idx])):
                tree[f"layers/{name}/vars/{i}"] = np.ascontiguousarray(a, dtype=np.float32)
        else:
            tree[f"layers/{name}/vars"] = {}
    tree["vars"] = {}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "3.3.3", "date_saved": datetime.now().strftime("%Y-%m-%d@%H:%M:%S"),
                                                "written_by": "cellscreen.model_io.cae_to_keras (no Keras install)"}))
        z.writestr("config.json", json.dumps(_keras_config(w, seq)))
        z.writestr("model.weights.h5", h5lite.write(tree))
    os.replace(tmp, path)


def has_native_files(model_dir: str) -> bool:
    return os.path.exists(os.path.join(model_dir, spec.NATIVE_CAE))


def has_reference_files(model_dir: str) -> bool:
    return all(os.path.exists(os.path.join(model_dir, f)) for f in spec.REF_MODEL_FILES)


def ensure_native_model_dir(model_dir: str) -> str:
    """What ProductionMutantScreening(model_dir) is handed may be the reference's six files (improved_detection.py:28-41)
    rather than the native set: convert them (in place; into a fresh temporary directory when model_dir is read-only)
    and return the directory cs_model_load should read.  A native set that is OLDER than the reference files it sits
    beside is refreshed."""
    native = has_native_files(model_dir)
    if not has_reference_files(model_dir):
        return model_dir                               # native only, or nothing: cs_model_load reports what is missing
    if native:
        newest_ref = max(os.path.getmtime(os.path.join(model_dir, f)) for f in spec.REF_MODEL_FILES)
        if os.path.getmtime(os.path.join(model_dir, spec.NATIVE_CAE)) >= newest_ref and \
                os.path.exists(os.path.join(model_dir, spec.NATIVE_DETECTOR)):
            return model_dir
    try:
        return convert_reference_model_dir(model_dir)
    except OSError:
        import tempfile
        return convert_reference_model_dir(model_dir, tempfile.mkdtemp(prefix="cellscreen_model_"))


def convert_reference_model_dir(model_dir: str, out_dir: Optional[str] = None) -> str:
    """The six files load_trained_models reads (improved_detection.py:28-41) -> the native file set
    (cae.bin, detector.bin, manifest.json) next to them (or in out_dir).  Needs scikit-learn for the pickles,
    nothing for the `.keras` archives."""
    ae = cae_from_keras(os.path.join(model_dir, "best_autoencoder.keras"))
    enc = cae_from_keras(os.path.join(model_dir, "encoder.keras"))
    det = detector_from_reference_pickles(model_dir)
    out = out_dir or model_dir
    save_model_dir(out, ae, enc, det)
    return out
