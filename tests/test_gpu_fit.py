"""Detector fit on the device (csrc/fit.hip, cs_fit_*) against the libraries the reference calls for it
(CAE_improved_modeltrain.py:408-427): scikit-learn's RobustScaler / PCA / OneClassSVM and, under them, numpy's
percentile and libsvm's solver.  Bars:

* scaler: bit-exact (center_ float32, scale_ float64) -- order statistics are exact and the interpolation
  arithmetic is numpy's, restated.
* PCA moments: mean_ bit-exact; the fp64 scatter matrix to 1e-12 of its largest entry; the principal axes
  against PCA(svd_solver='full') on the same scaled data.
* one-class SVM, points in general position: libsvm's own trajectory -- same iteration count, alpha to 1e-9, rho
  to 1e-9 relative (the solver's stopping tolerance is 1e-3; agreement this close means the same working sets were
  chosen all along).  Degenerate data (duplicates, 1-D): the solver's own tolerance, a few eps on rho and on the
  decision values.
"""
import os
import pickle

import numpy as np
import pytest

from cellscreen import detector_fit as df
from cellscreen import spec, synth
from cellscreen.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitter():
    f = df.Fitter(0)
    yield f
    f.close()


def _features(n, F=2048, seed=0):
    """Low-rank + noise through a ReLU: shaped like encoder features (non-negative, correlated, some dead columns)."""
    rng = np.random.default_rng(seed)
    z = rng.normal(size=(n, 24)).astype(np.float32)
    w = rng.normal(size=(24, F)).astype(np.float32)
    x = np.maximum(z @ w * 0.2 + rng.normal(size=(n, F)).astype(np.float32) * 0.3 + 0.2, 0).astype(np.float32)
    x[:, :4] = 0.0                                          # dead features: IQR 0 -> scale 1
    return x


@pytest.mark.parametrize("n", [1, 2, 3, 37, 599, 600, 1000, 1001, 4097])
def test_scaler_fit_is_bit_exact(fitter, n):
    from sklearn.preprocessing import RobustScaler
    rng = np.random.default_rng(n)
    F = 200                                                 # not a multiple of the transpose tile
    x = rng.normal(0.3, 1.0, (n, F)).astype(np.float32)
    x[:, 0] = 1.5                                           # constant
    x[:, 1] = np.round(x[:, 1] * 2) / 2                     # heavy ties
    x[:, 2] = -0.0
    x[::3, 3] = 0.0
    x[:, 4] *= 1e-30                                        # spread below 10 eps -> scale 1
    x[:, 5] = np.abs(x[:, 5]) * 1e30
    x[:, 6] = -np.abs(x[:, 6])
    want = RobustScaler().fit(x)
    center, scale = fitter.scaler(x)
    assert center.dtype == want.center_.dtype == np.float32 and scale.dtype == want.scale_.dtype == np.float64
    assert np.array_equal(center, want.center_)
    assert np.array_equal(scale, want.scale_)


def test_scaler_fit_rejects_nan_and_bad_arguments(fitter):
    x = np.zeros((10, 8), np.float32)
    x[3, 2] = np.nan
    with pytest.raises(RuntimeError, match="NaN"):
        fitter.scaler(x)
    with pytest.raises(RuntimeError):
        fitter.scaler(np.zeros((0, 8), np.float32))
    with pytest.raises(RuntimeError, match="multiple of 128"):
        fitter.pca_moments(np.zeros((10, 200), np.float32), np.zeros(200, np.float32), np.ones(200))
    with pytest.raises(RuntimeError, match="nu"):
        fitter.ocsvm(np.zeros((10, 3)), 1.0, 0.0)
    with pytest.raises(RuntimeError, match="n_components"):
        fitter.ocsvm(np.zeros((10, 129)), 1.0, 0.5)
    bad = np.zeros((10, 3))
    bad[4, 1] = np.inf
    with pytest.raises(RuntimeError, match="non-finite"):
        fitter.ocsvm(bad, 1.0, 0.5)


def test_scaler_fit_from_device_memory(fitter):
    import torch
    from sklearn.preprocessing import RobustScaler
    x = _features(2500, 512, seed=3)
    want = RobustScaler().fit(x)
    center, scale = fitter.scaler(torch.from_numpy(x).cuda())
    assert np.array_equal(center, want.center_) and np.array_equal(scale, want.scale_)


def test_pca_moments_and_axes(fitter):
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import RobustScaler
    x = _features(3001)
    sc = RobustScaler().fit(x)
    xs = sc.transform(x)
    mean, scatter = fitter.pca_moments(x, sc.center_, sc.scale_)
    assert np.array_equal(mean, xs.mean(axis=0))            # PCA.fit's mean_, float32 accumulation order included
    xc = (xs - xs.mean(axis=0)).astype(np.float64)          # PCA centres in float32, then factorises
    ref = xc.T @ xc
    assert np.abs(scatter - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.array_equal(scatter, scatter.T)
    k = 100
    comps, ev, total = df.principal_axes(scatter, x.shape[0], k)
    full = PCA(n_components=k, svd_solver="full").fit(xs)
    assert np.allclose(ev, full.explained_variance_, rtol=2e-4, atol=1e-6 * ev[0])          # sklearn factorises in float32
    assert abs(total - full.explained_variance_.sum() / full.explained_variance_ratio_.sum()) <= 2e-4 * total
    # the 24 planted directions are well separated from the noise floor: same axes, same signs
    lead = 24
    assert np.abs(np.abs(np.sum(comps[:lead] * full.components_[:lead], axis=1)) - 1).max() < 1e-3
    assert np.all(np.sum(comps[:lead] * full.components_[:lead], axis=1) > 0)
    # the rest spans the same subspace only approximately (flat spectrum); both bases are orthonormal
    assert np.abs(comps @ comps.T - np.eye(k)).max() < 1e-10


def _libsvm_pair(fitter, x, nu, **kw):
    from sklearn.svm import OneClassSVM
    want = OneClassSVM(kernel="rbf", gamma="scale", nu=nu, **kw).fit(x)
    got = fitter.ocsvm(x, want._gamma, nu, **kw)
    alpha = np.zeros(len(x))
    alpha[want.support_] = want.dual_coef_.ravel()
    return want, got, alpha


@pytest.mark.parametrize("n,d,nu", [(10, 3, 0.5), (257, 7, 0.999), (3000, 100, 0.05), (3000, 100, 0.10), (2000, 128, 0.3),
                                    (6000, 100, 0.05)])
@pytest.mark.parametrize("as_float32", [False, True])
def test_ocsvm_fit_follows_libsvm(fitter, n, d, nu, as_float32):
    """Points in general position: the device solver picks libsvm's working sets iteration for iteration.
    as_float32: values that are exactly floats (what PCA hands the SVM, CAE...:414-427) take the kernels that keep
    the training set as float32 in memory; the arithmetic is the same."""
    rng = np.random.default_rng(n * 1000 + d)
    x = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, d)
    if as_float32:
        x = x.astype(np.float32).astype(np.float64)
    want, got, alpha = _libsvm_pair(fitter, x, nu)
    assert got["status"] == 0 and want.fit_status_ == 0
    assert got["n_iter"] == want.n_iter_
    assert np.abs(got["alpha"] - alpha).max() <= 1e-9
    assert np.array_equal(np.flatnonzero(got["alpha"] > 0), want.support_)
    assert abs(got["rho"] + want.intercept_[0]) <= 1e-9 * max(1.0, abs(want.intercept_[0]))
    assert abs(got["alpha"].sum() - nu * n) <= 1e-9 * n     # the equality constraint of the dual
    # and the sklearn object rebuilt from the device solve answers like the fitted one
    rebuilt = df.sklearn_ocsvm(x, got["alpha"], got["rho"], want._gamma, nu, got["n_iter"])
    y = rng.normal(size=(64, d))
    assert np.abs(rebuilt.decision_function(y) - want.decision_function(y)).max() <= 1e-8 * max(1.0, abs(want.intercept_[0]))
    assert np.array_equal(rebuilt.predict(y), want.predict(y))


@pytest.mark.parametrize("n,d,nu,dup", [(500, 1, 0.1, 7), (800, 2, 0.2, 3), (1000, 5, 0.05, 2)])
def test_ocsvm_fit_degenerate_data_agrees_to_solver_tolerance(fitter, n, d, nu, dup):
    """Duplicated points and one- or two-dimensional data make many kernel values exactly or nearly equal; one float
    rounded the other way (libm vs ocml exp) then sends the two solvers down different but equally valid paths.  Both
    stop when the largest KKT violation is below eps = 1e-3 (svm.cpp:1029), which pins the gradient -- decision value
    plus rho -- to a few eps: that is the bar here.  The dual objective is compared as well."""
    rng = np.random.default_rng(n + d)
    x = rng.normal(size=(n, d))
    x[::dup] = x[0]
    want, got, alpha = _libsvm_pair(fitter, x, nu)
    eps = df.SVM_TOL
    assert got["status"] == 0
    assert abs(got["alpha"].sum() - nu * n) <= 1e-9 * n and got["alpha"].min() >= 0 and got["alpha"].max() <= 1
    assert abs(got["rho"] + want.intercept_[0]) <= 2 * eps
    rebuilt = df.sklearn_ocsvm(x, got["alpha"], got["rho"], want._gamma, nu, got["n_iter"])
    y = np.concatenate([x[:200], rng.normal(size=(100, d)) * 2])
    assert np.abs(rebuilt.decision_function(y) - want.decision_function(y)).max() <= 4 * eps
    k = np.exp(-want._gamma * ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    obj_ref = 0.5 * alpha @ k @ alpha
    assert abs(got["obj"] - obj_ref) <= eps * max(1.0, abs(obj_ref))


def test_ocsvm_max_iter_stops_like_libsvm(fitter):
    import warnings
    rng = np.random.default_rng(5)
    x = rng.normal(size=(1500, 20))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want, got, alpha = _libsvm_pair(fitter, x, 0.2, max_iter=40)
    assert got["status"] == 1 and got["n_iter"] == 40 == want.n_iter_
    assert np.abs(got["alpha"] - alpha).max() <= 1e-9
    assert abs(got["rho"] + want.intercept_[0]) <= 1e-9 * abs(want.intercept_[0])


def test_device_fit_end_to_end_scores_like_sklearn_objects(tmp_path):
    """fit_detector_device on encoder features -> (a) DetectorParams for cs_screen, (b) the reference's four
    pickles.  The reference's screening would unpickle (b) and call transform / decision_function
    (improved_detection.py:32-41, 134-142): cs_screen with (a) must agree with that, and both must agree with an
    all-scikit-learn fit of the same features up to the two fits' PCA difference."""
    w = synth.random_cae(seed=42)
    e0 = Engine.from_weights(w, None, None)
    feats = e0.encode(synth.blob_crops(7, 3000), which=0)
    e0.close()
    timings = {}
    det, objs = df.fit_detector_device(feats, output_dir=str(tmp_path), timings=timings)
    assert det.n_components == 100 and det.conservative.n_sv > 0 and det.moderate.n_sv > det.conservative.n_sv
    for name in ("scaler.pkl", "pca.pkl", "detector_conservative.pkl", "detector_moderate.pkl"):
        assert os.path.exists(tmp_path / name)
    test = synth.blob_crops(8, 600)
    test[::5] = synth.synth_crops(9, 0, 120)                                 # some crops unlike the training set
    e = Engine.from_weights(w, None, det)
    res = e.screen(test)
    tf = e.encode(test, which=0)
    e.close()
    with open(tmp_path / "scaler.pkl", "rb") as f:
        scaler = pickle.load(f)
    with open(tmp_path / "pca.pkl", "rb") as f:
        pca = pickle.load(f)
    with open(tmp_path / "detector_moderate.pkl", "rb") as f:
        moderate = pickle.load(f)
    red = pca.transform(scaler.transform(tf))                                # :134-135 on the unpickled objects
    dec = moderate.decision_function(red)                                    # :142
    scale = max(1.0, np.abs(dec).max())
    assert np.abs(-res["mod_score"] - dec).max() <= 1e-4 * scale       # float32 PCA on both sides, different sum order
    # against the host scikit-learn fit (randomized PCA, seeded): same flags except near the boundary
    det_sk, objs_sk = df.fit_detector(feats, pca_random_state=0)
    assert np.array_equal(det.scaler_center, det_sk.scaler_center) and np.array_equal(det.scaler_scale, det_sk.scaler_scale)
    red_sk = objs_sk["pca"].transform(objs_sk["scaler"].transform(tf))
    for name, key in (("Conservative", "cons"), ("Moderate", "mod")):
        d_sk = objs_sk["detectors"][name].decision_function(red_sk)
        mine = -res[f"{key}_score"]
        rate_sk, rate = (d_sk < 0).mean(), (mine < 0).mean()
        agree = ((d_sk < 0) == (mine < 0)).mean()
        assert agree >= 0.97, f"{name}: flag agreement {agree:.3f}"
        assert abs(rate - rate_sk) <= 0.03, f"{name}: anomaly rate {rate:.3f} vs {rate_sk:.3f}"
        assert np.corrcoef(d_sk, mine)[0, 1] > 0.99
    # baseline anomaly rates on the training set (:430-434) sit at nu, as libsvm guarantees
    for name, nu in (("Conservative", spec.NU_CONSERVATIVE), ("Moderate", spec.NU_MODERATE)):
        pred = objs["detectors"][name].predict(objs["features_reduced"])
        assert abs((pred == -1).mean() - nu) < 0.02
    assert all(k in timings for k in ("scaler_s", "pca_moments_s", "pca_eigh_s", "svm_moderate_iter"))


def test_ocsvm_fit_at_twenty_thousand_points(fitter):
    """A training set of the reference's order of magnitude (libsvm needs a few seconds for it)."""
    rng = np.random.default_rng(11)
    z = rng.normal(size=(20000, 12))
    x = (z @ rng.normal(size=(12, 100)) * 0.3 + rng.normal(size=(20000, 100))).astype(np.float32).astype(np.float64)
    want, got, alpha = _libsvm_pair(fitter, x, 0.05)
    eps = df.SVM_TOL
    assert got["status"] == 0
    assert abs(got["rho"] + want.intercept_[0]) <= 2 * eps                     # the solver-tolerance bar always holds
    assert abs(got["alpha"].sum() - 0.05 * len(x)) <= 1e-9 * len(x)
    same_path = got["n_iter"] == want.n_iter_
    if same_path:                                                              # and normally the path is libsvm's own
        assert np.abs(got["alpha"] - alpha).max() <= 1e-9
    else:
        sv_got, sv_want = set(np.flatnonzero(got["alpha"] > 0)), set(want.support_)
        assert len(sv_got ^ sv_want) <= 0.02 * len(sv_want)
    print(f"n_iter {got['n_iter']} vs libsvm {want.n_iter_}; max |d alpha| {np.abs(got['alpha'] - alpha).max():.3g}")


def test_ocsvm_solution_satisfies_kkt_at_reference_scale(fitter):
    """Size-independent check at the reference's training-set size (50,000 x 100; libsvm needs ~20 s there): the
    returned alpha must be dual feasible and pass libsvm's own stopping test (svm.cpp:1029) when the gradient
    G = Q alpha is recomputed from scratch in numpy float64 -- independent of the solver's incrementally updated G."""
    rng = np.random.default_rng(21)
    n, d, nu = 50000, 100, 0.05
    z = rng.normal(size=(n, 10))
    x = (z @ rng.normal(size=(10, d)) * 0.4 + rng.normal(size=(n, d))).astype(np.float32).astype(np.float64)
    gamma = 1.0 / (d * x.var())
    got = fitter.ocsvm(x, gamma, nu)
    alpha = got["alpha"]
    assert got["status"] == 0 and got["n_iter"] > 0
    assert alpha.min() >= 0.0 and alpha.max() <= 1.0 and abs(alpha.sum() - nu * n) <= 1e-9 * n
    sv = np.flatnonzero(alpha > 0)
    xs, a = x[sv], alpha[sv]
    sq, sqs = (x * x).sum(1), (xs * xs).sum(1)
    g = np.empty(n)
    for lo in range(0, n, 4096):
        hi = min(n, lo + 4096)
        k = np.exp(-gamma * np.maximum(sq[lo:hi, None] + sqs[None, :] - 2.0 * (x[lo:hi] @ xs.T), 0.0))
        g[lo:hi] = k.astype(np.float32).astype(np.float64) @ a          # Qfloat rounding, as the solver sees Q
    gmax = (-g[alpha < 1.0]).max()                                       # i-candidates: not at the upper bound
    gmax2 = g[alpha > 0.0].max()                                         # j-candidates: not at the lower bound
    assert gmax + gmax2 < df.SVM_TOL * 1.01, f"KKT gap {gmax + gmax2:.3e}"
    free = (alpha > 0) & (alpha < 1)
    assert free.any() and abs(g[free].mean() - got["rho"]) <= 1e-6 * abs(got["rho"])
    assert abs(0.5 * (alpha @ g) - got["obj"]) <= 1e-6 * abs(got["obj"])
    # nu-property: at most nu*n points outside (alpha = 1 are the margin errors), at least nu*n support vectors
    assert (alpha >= 1.0).sum() <= nu * n <= len(sv)


def test_device_fit_with_fewer_cells_than_components():
    """n_components = min(100, F, N - 1) (:412): 60 training cells give 59 components; scikit-learn's PCA picks its exact
    'full' solver for that shape, so both fits are deterministic and can be compared directly."""
    w = synth.random_cae(seed=42)
    e0 = Engine.from_weights(w, None, None)
    feats = e0.encode(synth.blob_crops(3, 60), which=0)
    test = e0.encode(synth.blob_crops(4, 200), which=0)
    e0.close()
    det, objs = df.fit_detector_device(feats)
    det_sk, objs_sk = df.fit_detector(feats)
    assert det.n_components == det_sk.n_components == 59
    assert objs_sk["pca"]._fit_svd_solver == "full"
    assert np.array_equal(det.scaler_center, det_sk.scaler_center) and np.array_equal(det.scaler_scale, det_sk.scaler_scale)
    assert np.array_equal(det.pca_mean, det_sk.pca_mean)
    ev, ev_sk = objs["pca"].explained_variance_, objs_sk["pca"].explained_variance_
    assert np.allclose(ev, ev_sk, rtol=1e-3, atol=1e-5 * ev_sk[0])
    lead = int(np.sum(ev_sk > 1e-3 * ev_sk[0]))                          # axes with a spectral gap worth the name
    dots = np.sum(det.pca_components[:lead].astype(np.float64) * det_sk.pca_components[:lead], axis=1)
    assert np.all(dots > 0.98), dots.min()                               # same axes, same signs
    for name in ("Conservative", "Moderate"):
        mine = objs["detectors"][name].decision_function(objs["pca"].transform(objs["scaler"].transform(test)))
        ref = objs_sk["detectors"][name].decision_function(objs_sk["pca"].transform(objs_sk["scaler"].transform(test)))
        assert np.corrcoef(mine, ref)[0, 1] > 0.995
        assert ((mine < 0) == (ref < 0)).mean() >= 0.97


def test_scaler_fit_with_millions_of_rows(fitter):
    """More than 65,535 row tiles (the grid.y limit the transpose kernel must not depend on)."""
    from sklearn.preprocessing import RobustScaler
    rng = np.random.default_rng(77)
    x = rng.gamma(2.0, 1.0, (2_200_000, 8)).astype(np.float32)
    want = RobustScaler().fit(x)
    center, scale = fitter.scaler(x)
    assert np.array_equal(center, want.center_) and np.array_equal(scale, want.scale_)
