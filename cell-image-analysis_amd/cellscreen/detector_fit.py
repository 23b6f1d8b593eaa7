"""create_anomaly_detector (CAE_improved_modeltrain.py:394-446): encoder features ->
RobustScaler -> PCA -> two OneClassSVMs.

Two implementations of the same fit:

* `fit_detector_device` -- on the GPU through the C ABI (csrc/fit.hip): the per-feature order
  statistics, the scaled/centred moments, the projection and the whole SMO iteration run on the
  device; the host does numpy's interpolation arithmetic on the selected order statistics, the
  F x F symmetric eigenproblem (LAPACK through scipy) and libsvm's final rho sum.  No CPU route
  for the data-proportional work.
* `fit_detector` -- scikit-learn on the host, the library the reference itself calls; kept as the
  explicit alternative (`method="sklearn"` in the training mirror) and as the oracle of the
  device fit in the tests.

Both return the arrays the device scoring path consumes plus scikit-learn objects, so the
reference's four pickles (:437-444) can be written either way."""
from __future__ import annotations

import ctypes as C
import os
import pickle
import time
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib as L
from . import spec
from .model_io import detector_params_from_sklearn
from .spec import DetectorParams, OCSVMParams

MEM_HOST, MEM_DEVICE = 0, 1
SVM_TOL = 1e-3              # OneClassSVM(tol=1e-3) default -> libsvm eps
EIGH_THREADS = 16           # BLAS threads for the F x F eigenproblem (measured: 0.18 s at 8-16, 0.79 s at 256)


def _write_pickles(output_dir, scaler, pca, detectors):                # :437-444
    os.makedirs(output_dir, exist_ok=True)
    for name, obj in (("scaler.pkl", scaler), ("pca.pkl", pca),
                      ("detector_conservative.pkl", detectors["Conservative"]),
                      ("detector_moderate.pkl", detectors["Moderate"])):
        with open(os.path.join(output_dir, name), "wb") as f:
            pickle.dump(obj, f)


def fit_detector(features_flat: np.ndarray, output_dir: Optional[str] = None, pca_random_state=None):
    """scikit-learn on the host.  features_flat: (N, F) float32 encoder features flattened (h,w,c)
    (:401-402).  Returns (DetectorParams, dict(scaler=..., pca=..., detectors={'Conservative':..,
    'Moderate':..})).  When output_dir is given, writes the reference's four pickles (:437-444)."""
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
    if output_dir is not None:
        _write_pickles(output_dir, scaler, pca, detectors)
    params = detector_params_from_sklearn(scaler, pca, detectors["Conservative"], detectors["Moderate"])
    return params, dict(scaler=scaler, pca=pca, detectors=detectors, features_reduced=features_reduced)


# ---- device fit ---------------------------------------------------------------------------------
class Fitter:
    """One cs_fit handle (one GPU, one stream).  `features` arguments: a C-contiguous float32 numpy
    array (N, F), or a contiguous CUDA float32 torch tensor of that shape (stays on the device)."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load_library()
        self._h = C.c_void_p()
        L.check(self._lib.cs_fit_create(device_id, C.byref(self._h)))
        self.device_id = device_id

    def close(self):
        if self._h:
            self._lib.cs_fit_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def last_ms(self) -> float:
        v = C.c_double()
        L.check(self._lib.cs_fit_last_ms(self._h, C.byref(v)))
        return v.value

    def _features(self, features):
        if isinstance(features, np.ndarray):
            if features.ndim != 2:
                raise ValueError(f"features must be (N, F), got {features.shape}")
            f = np.ascontiguousarray(features, np.float32)
            return f, f.ctypes.data, f.shape[0], f.shape[1], MEM_HOST
        import torch
        if not (features.is_cuda and features.dtype == torch.float32 and features.is_contiguous() and features.dim() == 2):
            raise ValueError("device features must be a contiguous CUDA float32 tensor (N, F)")
        L.order_after_torch(self._lib.cs_fit_wait_stream, self._h, features)
        return features, features.data_ptr(), features.shape[0], features.shape[1], MEM_DEVICE

    def scaler(self, features) -> Tuple[np.ndarray, np.ndarray]:
        """RobustScaler().fit (:408): (center_ float32 (F,), scale_ float64 (F,))."""
        keep, ptr, n, F, kind = self._features(features)
        center = np.empty(F, np.float32)
        scale = np.empty(F, np.float64)
        L.check(self._lib.cs_fit_scaler(self._h, ptr, n, F, kind, center.ctypes.data, scale.ctypes.data))
        return center, scale

    def pca_moments(self, features, center: np.ndarray, scale: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(mean_ float32 (F,), scatter float64 (F, F)) of the scaled features."""
        keep, ptr, n, F, kind = self._features(features)
        center = np.ascontiguousarray(center, np.float32)
        scale = np.ascontiguousarray(scale, np.float64)
        mean = np.empty(F, np.float32)
        scatter = np.empty((F, F), np.float64)
        L.check(self._lib.cs_fit_pca_moments(self._h, ptr, n, F, kind, center.ctypes.data, scale.ctypes.data,
                                             mean.ctypes.data, scatter.ctypes.data))
        return mean, scatter

    def project(self, features, center, scale, components, mean_proj) -> np.ndarray:
        """pca.transform(scaler.transform(features)) with the given parameters -> (N, C) float32."""
        keep, ptr, n, F, kind = self._features(features)
        center = np.ascontiguousarray(center, np.float32)
        scale = np.ascontiguousarray(scale, np.float64)
        components = np.ascontiguousarray(components, np.float32)
        mean_proj = np.ascontiguousarray(mean_proj, np.float32)
        Cn = components.shape[0]
        if components.shape != (Cn, F) or mean_proj.shape != (Cn,):
            raise ValueError("components must be (C, F) and mean_proj (C,)")
        out = np.empty((n, Cn), np.float32)
        L.check(self._lib.cs_fit_project(self._h, ptr, n, F, kind, center.ctypes.data, scale.ctypes.data,
                                         components.ctypes.data, mean_proj.ctypes.data, Cn, out.ctypes.data))
        return out

    def ocsvm(self, x: np.ndarray, gamma: float, nu: float, tol: float = SVM_TOL, max_iter: int = -1) -> Dict:
        """OneClassSVM(kernel='rbf', gamma=gamma, nu=nu, tol=tol, max_iter=max_iter).fit(x) as libsvm
        solves it.  x: (N, D) float64.  Returns alpha (N,), rho, obj, n_iter, status (0 converged,
        1 stopped at max_iter)."""
        x = np.ascontiguousarray(x, np.float64)
        if x.ndim != 2:
            raise ValueError("x must be (N, D)")
        n, d = x.shape
        alpha = np.empty(n, np.float64)
        rho, obj = C.c_double(), C.c_double()
        n_iter, status = C.c_int64(), C.c_int32()
        L.check(self._lib.cs_fit_ocsvm(self._h, x.ctypes.data, n, d, float(gamma), float(nu), float(tol), int(max_iter),
                                       alpha.ctypes.data, C.byref(rho), C.byref(obj), C.byref(n_iter), C.byref(status)))
        return dict(alpha=alpha, rho=rho.value, obj=obj.value, n_iter=n_iter.value, status=status.value)


def principal_axes(scatter: np.ndarray, n_samples: int, n_components: int):
    """The leading eigenpairs of scatter / (n - 1), ordered and signed as PCA.fit leaves components_
    (descending variance; svd_flip(u_based_decision=False): the entry of largest magnitude of every
    row is positive, sklearn/utils/extmath.py:944-952).  Returns (components float64 (C, F),
    explained_variance (C,), total_variance)."""
    from scipy.linalg import eigh
    F = scatter.shape[0]
    cov = scatter / max(n_samples - 1, 1)
    try:        # LAPACK's tridiagonalisation stops scaling at a few cores; on a 256-thread host the default pool is 4x slower
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=max(1, min(EIGH_THREADS, os.cpu_count() or 1)))   # never above the pool's initial size
    except ImportError:
        import contextlib
        limit = contextlib.nullcontext()
    with limit:
        w, v = eigh(cov, subset_by_index=[F - n_components, F - 1])
    w, v = w[::-1], v[:, ::-1]
    comps = np.ascontiguousarray(v.T)
    idx = np.argmax(np.abs(comps), axis=1)
    comps *= np.sign(comps[np.arange(n_components), idx])[:, None]
    return comps, np.maximum(w, 0.0), float(np.trace(cov))


def _sklearn_scaler(center, scale):
    from sklearn.preprocessing import RobustScaler
    s = RobustScaler()
    s.center_, s.scale_, s.n_features_in_ = center, scale, center.shape[0]
    return s


def _sklearn_pca(components, mean, explained_variance, total_var, n_samples):
    from sklearn.decomposition import PCA
    k, F = components.shape
    p = PCA(n_components=k)
    p.components_, p.mean_ = components, mean
    p.n_components_, p.n_features_in_, p.n_samples_ = k, F, n_samples
    p.explained_variance_ = explained_variance
    p.explained_variance_ratio_ = explained_variance / total_var if total_var > 0 else np.zeros_like(explained_variance)
    p.singular_values_ = np.sqrt(explained_variance * max(n_samples - 1, 1))
    rest = min(F, n_samples) - k
    p.noise_variance_ = float((total_var - explained_variance.sum()) / rest) if rest > 0 else 0.0
    p._fit_svd_solver = "covariance_eigh"
    return p


def sklearn_ocsvm(x: np.ndarray, alpha: np.ndarray, rho: float, gamma: float, nu: float, n_iter: int = 0, status: int = 0):
    """A OneClassSVM carrying a finished solve (what BaseLibSVM.fit stores, sklearn/svm/_base.py:255-286,
    _classes.py:1733-1737), so the reference's `pickle.load(...).decision_function` works on it."""
    from sklearn.svm import OneClassSVM
    sv = np.flatnonzero(alpha > 0).astype(np.int32)
    d = OneClassSVM(kernel="rbf", gamma="scale", nu=nu)
    d._sparse, d._gamma = False, float(gamma)
    d.support_ = sv
    d.support_vectors_ = np.ascontiguousarray(x[sv], np.float64)
    d._n_support = np.array([sv.size, sv.size], np.int32)   # what sklearn's wrapper reports for a one-class model
    d.dual_coef_ = np.ascontiguousarray(alpha[sv], np.float64).reshape(1, -1)
    d.intercept_ = np.array([-rho], np.float64)
    d._dual_coef_, d._intercept_ = d.dual_coef_, d.intercept_.copy()
    d._probA, d._probB = np.empty(0, np.float64), np.empty(0, np.float64)
    d.fit_status_, d.shape_fit_ = int(status), x.shape
    d.n_features_in_ = x.shape[1]
    d._num_iter = np.array([n_iter], np.int32)
    d.n_iter_ = int(n_iter)
    d.offset_ = -d._intercept_
    return d


def fit_detector_device(features_flat, output_dir: Optional[str] = None, device_id: int = 0, timings: Optional[dict] = None):
    """The same fit with the data-proportional work on the GPU (csrc/fit.hip).  PCA: the exact
    principal axes (eigenvectors of the covariance) where the reference's PCA(svd_solver='auto')
    takes the randomized solver with an unseeded generator (:413) -- the two agree to the
    randomized solver's accuracy, see DESIGN.md section 3f."""
    t = timings if timings is not None else {}
    with Fitter(device_id) as fit:
        n, F = int(features_flat.shape[0]), int(features_flat.shape[1])
        t0 = time.perf_counter()
        center, scale = fit.scaler(features_flat)                                         # :408
        t["scaler_s"], t["scaler_device_ms"] = time.perf_counter() - t0, fit.last_ms
        n_components = min(spec.PCA_MAX_COMPONENTS, F, n - 1)                              # :412
        t0 = time.perf_counter()
        mean, scatter = fit.pca_moments(features_flat, center, scale)                      # :413-414
        t["pca_moments_s"], t["pca_moments_device_ms"] = time.perf_counter() - t0, fit.last_ms
        t0 = time.perf_counter()
        comps64, ev, total_var = principal_axes(scatter, n, n_components)
        components = comps64.astype(np.float32)
        mean_proj = (mean.reshape(1, -1) @ components.T).ravel()                           # sklearn _base.py:152-153
        t["pca_eigh_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        reduced = fit.project(features_flat, center, scale, components, mean_proj)
        t["project_s"] = time.perf_counter() - t0
        x64 = reduced.astype(np.float64)                                                   # sklearn svm/_base.py:190 (dtype=np.float64)
        x_var = x64.var()
        gamma = 1.0 / (x64.shape[1] * x_var) if x_var != 0 else 1.0                        # svm/_base.py:244-247
        solved = {}
        for name, nu in (("Conservative", spec.NU_CONSERVATIVE), ("Moderate", spec.NU_MODERATE)):   # :420-427
            t0 = time.perf_counter()
            solved[name] = fit.ocsvm(x64, gamma, nu)
            t[f"svm_{name.lower()}_s"], t[f"svm_{name.lower()}_device_ms"] = time.perf_counter() - t0, fit.last_ms
            t[f"svm_{name.lower()}_iter"] = solved[name]["n_iter"]

    def params(r):
        sv = r["alpha"] > 0
        return OCSVMParams(np.ascontiguousarray(x64[sv]), np.ascontiguousarray(r["alpha"][sv]), float(gamma), float(r["rho"]))
    det = DetectorParams(center, scale, components, mean, mean_proj.astype(np.float32),
                         params(solved["Conservative"]), params(solved["Moderate"]))
    objs = dict(scaler=_sklearn_scaler(center, scale),
                pca=_sklearn_pca(components, mean, ev, total_var, n),
                detectors={k: sklearn_ocsvm(x64, r["alpha"], r["rho"], gamma, nu, r["n_iter"], r["status"])
                           for (k, r), nu in zip(solved.items(), (spec.NU_CONSERVATIVE, spec.NU_MODERATE))},
                features_reduced=reduced, solved=solved)
    if output_dir is not None:
        _write_pickles(output_dir, objs["scaler"], objs["pca"], objs["detectors"])
    return det, objs
