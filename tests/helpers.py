"""Shared test helpers: golden fixtures -> parameter containers, tolerance checks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from cellscreen.spec import CAEWeights, DetectorParams, OCSVMParams  # noqa: E402


def cae_from_golden(g) -> CAEWeights:
    n_conv = 7
    return CAEWeights([g[f"conv{l}_kernel"] for l in range(n_conv)], [g[f"conv{l}_bias"] for l in range(n_conv)],
                      [g[f"bn{l}_gamma"] for l in range(n_conv - 1)], [g[f"bn{l}_beta"] for l in range(n_conv - 1)],
                      [g[f"bn{l}_mean"] for l in range(n_conv - 1)], [g[f"bn{l}_var"] for l in range(n_conv - 1)],
                      bn_eps=float(g["bn_eps"])).validate()


def det_from_golden(g) -> DetectorParams:
    return DetectorParams(g["scaler_center"], g["scaler_scale"], g["pca_components"], g["pca_mean"], g["pca_mean_proj"],
                          OCSVMParams(g["cons_sv"], g["cons_coef"], float(g["cons_gamma"]), float(g["cons_rho"])),
                          OCSVMParams(g["mod_sv"], g["mod_coef"], float(g["mod_gamma"]), float(g["mod_rho"])))


# ---- stated tolerances, all measured against an fp64-evaluated reference
TOL_FEATURES = 1e-5      # max abs <= 1e-5 * max|feature|
TOL_RECON = 1e-5         # max abs
TOL_ERR_REL = 1e-5       # per-cell MSE / MAE, relative
TOL_STAGE = 1e-5         # scaled / PCA outputs: max abs <= 1e-5 * max|output| with oracle inputs
TOL_DEC_STAGE = 1e-9     # decision values with oracle PCA inputs: abs <= 1e-9 * sum|alpha|
TOL_DEC_E2E = 1e-4       # end to end: abs <= 1e-4 * sum|alpha|


def assert_close_scaled(got, ref, tol, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:g} * {scale:.3e}"
    return err / scale


def assert_rel(got, ref, tol, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    assert rel.max() <= tol, f"{what}: max rel err {rel.max():.3e} > {tol:g}"
    return rel.max()


def flags_agree(dec_got, pred_got, dec_ref, pred_ref, tol_abs, what):
    """Labels must be identical wherever |dec| exceeds the score tolerance; returns #skipped."""
    dec_ref = np.asarray(dec_ref)
    sure = np.abs(dec_ref) > tol_abs
    assert np.array_equal(np.asarray(pred_got)[sure], np.asarray(pred_ref)[sure]), f"{what}: label mismatch away from 0"
    return int((~sure).sum())
