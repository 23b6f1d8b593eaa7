"""Exploratory check of the device detector fit against scikit-learn (run on a GPU box)."""
import sys, time, json
import numpy as np
sys.path.insert(0, "cell-image-analysis_amd")
from cellscreen import detector_fit as df
from sklearn.preprocessing import RobustScaler
from sklearn.svm import OneClassSVM

rng = np.random.default_rng(0)
fit = df.Fitter(0)
# scaler
for n in (1, 2, 37, 599, 600, 1000, 1001):
    F = 256
    X = rng.normal(0.3, 1.0, (n, F)).astype(np.float32)
    X[:, 0] = 1.5
    X[:, 1] = np.round(X[:, 1] * 2) / 2
    X[:, 2] = -0.0
    if n > 2: X[::3, 3] = 0.0
    sc = RobustScaler().fit(X)
    c, s = fit.scaler(X)
    print("scaler", n, np.array_equal(c, sc.center_), np.array_equal(s, sc.scale_), c.dtype, sc.center_.dtype, sc.scale_.dtype, flush=True)
    if not np.array_equal(s, sc.scale_):
        bad = np.flatnonzero(s != sc.scale_); print(bad[:5], s[bad[:5]], sc.scale_[bad[:5]])
    if not np.array_equal(c, sc.center_):
        bad = np.flatnonzero(c != sc.center_); print(bad[:5], c[bad[:5]], sc.center_[bad[:5]])
# moments
n, F = 3000, 2048
Z = rng.normal(size=(n, 40)).astype(np.float32); W = rng.normal(size=(40, F)).astype(np.float32)
X = np.maximum(Z @ W * 0.2 + rng.normal(size=(n, F)).astype(np.float32) * 0.3 + 0.2, 0).astype(np.float32)
sc = RobustScaler().fit(X)
c, s = fit.scaler(X)
print("scaler big", np.array_equal(c, sc.center_), np.array_equal(s, sc.scale_), fit.last_ms)
Xs = sc.transform(X)
mean, scat = fit.pca_moments(X, c, s)
print("mean exact", np.array_equal(mean, Xs.mean(axis=0)), "ms", fit.last_ms)
Xc = (Xs - Xs.mean(axis=0)).astype(np.float64)
ref = Xc.T @ Xc
print("scatter rel err", np.abs(scat - ref).max() / np.abs(ref).max(), "sym", np.abs(scat - scat.T).max())
# svm
for (n, D, nu) in ((10, 3, 0.5), (500, 1, 0.1), (3000, 100, 0.05), (3000, 100, 0.10), (2000, 128, 0.3)):
    X = rng.normal(size=(n, D)) * rng.uniform(0.5, 2.0, D)
    if n == 500: X[::7] = X[0]
    t0 = time.time(); d = OneClassSVM(kernel="rbf", gamma="scale", nu=nu).fit(X); ts = time.time() - t0
    t0 = time.time(); r = fit.ocsvm(X, d._gamma, nu); tg = time.time() - t0
    a = np.zeros(n); a[d.support_] = d.dual_coef_.ravel()
    print(f"svm n={n} D={D} nu={nu}: iters {d.n_iter_} vs {r['n_iter']}; rho {-d.intercept_[0]:.12g} vs {r['rho']:.12g}; "
          f"max|dalpha| {np.abs(a - r['alpha']).max():.3g}; nsv {d.support_.size} vs {(r['alpha'] > 0).sum()}; "
          f"sklearn {ts:.3f}s device {tg:.3f}s ({fit.last_ms:.1f} ms)", flush=True)
