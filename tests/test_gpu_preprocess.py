"""Parity of the HIP crop preprocess (csrc/preprocess.hip, through cs_preprocess) against the real
scikit-image/SciPy outputs in tests/golden/golden_preprocess.npz and against the CPU oracle on
seeded ragged crops.  CLAHE is integer work: bit-exact.  The fp64 resize is cast to float32 at the
boundary (improved_detection.py:122): tolerance one float32 rounding, 6e-8 absolute on [0,1]."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cellscreen import _lib as L
from cellscreen import preprocess as pp
from cellscreen import synth
from oracle import preprocess_oracle as po

pytestmark = pytest.mark.gpu

TOL_OUT = 6e-8          # |fp32(hip fp64 result) - reference fp64 result|: one float32 rounding below 1.0


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "golden_preprocess.npz"))


@pytest.fixture(scope="module")
def proc():
    p = pp.Preprocessor(0)
    yield p
    p.close()


def _run(proc, crops, clip=0.02):
    pix, off, hs, ws = pp.pack_crops(crops)
    out, cl = proc.run_packed(pix, off, hs, ws, clip, want_clahe=True)
    return out, pp.split_clahe(cl, off, hs, ws)


@pytest.mark.parametrize("dt", [np.uint8, np.uint16])
def test_golden_vectors(proc, gold, dt):
    idx = [i for i in range(int(gold["n"])) if gold[f"crop_{i}"].dtype == dt]
    out, cl = _run(proc, [gold[f"crop_{i}"] for i in idx], float(gold["clip_limit"]))
    assert out.dtype == np.float32 and out.shape == (len(idx), 64, 64)
    for k, i in enumerate(idx):
        assert np.array_equal(cl[k], gold[f"clahe_u16_{i}"]), f"CLAHE stage of golden crop {i} is not bit-exact"
        err = np.abs(out[k].astype(np.float64) - gold[f"out_{i}"]).max()
        assert err <= TOL_OUT, f"golden crop {i} {gold[f'crop_{i}'].shape}: {err:.3e}"


@pytest.mark.parametrize("dt,seed", [(np.uint8, 11), (np.uint16, 12)])
def test_seeded_ragged_crops_against_oracle(proc, dt, seed):
    crops = synth.raw_crops(seed, 96, dt, 8, 150, flat_every=5)
    out, cl = _run(proc, crops)
    for k, c in enumerate(crops):
        assert np.array_equal(cl[k], po.clahe_u16(c)), f"crop {k} {c.shape}"
        err = np.abs(out[k].astype(np.float64) - po.preprocess_crop(c)).max()
        assert err <= TOL_OUT, f"crop {k} {c.shape}: {err:.3e}"


def test_clip_limit_variants(proc):
    crops = synth.raw_crops(21, 12, np.uint8, 16, 90)
    for clip in (0.0, 0.005, 0.1, 1.0):              # 0 and >= 1: plain AHE (_adapthist.py:118-119)
        out, cl = _run(proc, crops, clip)
        for k, c in enumerate(crops):
            assert np.array_equal(cl[k], po.clahe_u16(c, clip)), f"clip {clip} crop {k}"


def test_extreme_shapes_and_values(proc):
    rng = np.random.default_rng(5)
    crops = [np.full((8, 8), 200, np.uint8),                         # constant, smallest legal
             np.zeros((40, 33), np.uint8),                           # all zero
             np.full((17, 90), 255, np.uint8),                       # saturated
             rng.integers(0, 2, (15, 15)).astype(np.uint8),          # two grey levels, k = 1 tiles (15x15 of them)
             rng.integers(0, 256, (8, 300)).astype(np.uint8),        # strong anisotropic down-scale
             rng.integers(0, 256, (300, 9)).astype(np.uint8),
             rng.integers(0, 256, (257, 255)).astype(np.uint8)]
    out, cl = _run(proc, crops)
    for k, c in enumerate(crops):
        assert np.array_equal(cl[k], po.clahe_u16(c)), f"crop {k} {c.shape}"
        assert np.abs(out[k].astype(np.float64) - po.preprocess_crop(c)).max() <= TOL_OUT, f"crop {k} {c.shape}"
    assert np.isfinite(out).all() and out.min() >= 0.0 and out.max() <= 1.0


def test_identity_resize_property_at_64(proc):
    """At 64x64 the resize is the identity, so out == float32(rescaled CLAHE image) exactly --
    checked on 4096 crops without the oracle (size-independent property)."""
    crops = synth.raw_crops(31, 64, np.uint16, 64, 64) * 64
    out, cl = _run(proc, crops)
    for k in range(0, len(crops), 257):
        assert np.array_equal(cl[k], po.clahe_u16(crops[k]))
    cl = np.stack(cl).astype(np.float64) * (1.0 / 65535.0)
    lo, hi = cl.min(axis=(1, 2), keepdims=True), cl.max(axis=(1, 2), keepdims=True)
    eq = (cl - lo) / (hi - lo)
    assert np.abs(out.astype(np.float64) - eq).max() <= TOL_OUT
    assert np.array_equal(out[:64], out[64:128])                     # same input, same bits


def test_empty_and_bad_arguments(proc):
    lib = L.load_library()
    assert proc([]).shape == (0, 64, 64)
    with pytest.raises(RuntimeError, match="below 8 px"):
        proc([np.zeros((7, 30), np.uint8)])
    with pytest.raises(RuntimeError, match="above 1024"):
        proc([np.zeros((1025, 8), np.uint8)])
    pix = np.zeros(200, np.uint8)
    with pytest.raises(RuntimeError, match="overlaps"):
        proc.run_packed(pix, np.array([0, 50]), np.array([10, 10]), np.array([10, 10]))
    with pytest.raises(RuntimeError, match="outside the pixel buffer"):
        proc.run_packed(pix, np.array([150]), np.array([10]), np.array([10]))
    assert lib.cs_preprocess(None, None, 0, 0, 0, None, None, None, 1, 0.02, None, None, 0) == -1


def test_device_resident_in_and_out_feeds_the_screening_path(proc):
    """pixels and the [n,64,64] result stay in HBM; the result is what cs_screen consumes."""
    import torch
    from cellscreen.engine import Engine
    crops = synth.raw_crops(41, 40, np.uint8, 20, 100)
    pix, off, hs, ws = pp.pack_crops(crops)
    host = proc.run_packed(pix, off, hs, ws)
    d_pix = torch.from_numpy(pix).cuda()
    d_out = torch.empty((len(crops), 64, 64), dtype=torch.float32, device="cuda")
    proc.run_packed(d_pix, off, hs, ws, out=d_out)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), host)
    e = Engine.from_weights(synth.random_cae(seed=42))
    try:
        _, mse_dev, _ = e.reconstruct(d_out, want_recon=False)
        _, mse_host, _ = e.reconstruct(host, want_recon=False)
        assert np.array_equal(mse_dev.cpu().numpy(), np.asarray(mse_host))
    finally:
        e.close()
    ms, px = proc.last_timing()
    assert ms > 0 and px == pix.size


def test_raw_crops_to_flags_end_to_end(proc):
    """improved_detection.py:98-99 -> :117-153 on the device (crops stay in HBM between cs_preprocess and
    cs_screen) against the CPU chain preprocess oracle -> CAE/detector oracle."""
    import torch
    import helpers as H
    from cellscreen.detector_fit import fit_detector
    from cellscreen.engine import Engine
    from oracle import oracle
    w = synth.random_cae(seed=42)
    raw_fit = synth.raw_crops(51, 400, np.uint16, 24, 90)
    raw = synth.raw_crops(52, 48, np.uint16, 16, 120, flat_every=7)
    e0 = Engine.from_weights(w)
    det, _ = fit_detector(e0.encode(proc(raw_fit), which=0), pca_random_state=0)
    e0.close()
    pix, off, hs, ws = pp.pack_crops(raw)
    d_out = torch.empty((len(raw), 64, 64), dtype=torch.float32, device="cuda")
    proc.run_packed(torch.from_numpy(pix.view(np.int16)).cuda(), off, hs, ws, out=d_out)
    e = Engine.from_weights(w, None, det)
    try:
        r = {k: v.cpu().numpy() for k, v in e.screen(d_out).items()}
    finally:
        e.close()
    x_ref = po.preprocess_crops(raw)
    assert np.abs(d_out.cpu().numpy() - x_ref).max() <= TOL_OUT
    ref = oracle.screen(w, None, det, x_ref, acc64=True)
    H.assert_rel(r["mse"], ref["mse"], H.TOL_ERR_REL, "mse")
    for name, p in (("cons", det.conservative), ("mod", det.moderate)):
        tol = H.TOL_DEC_E2E * np.abs(p.dual_coef).sum()
        assert np.abs(r[f"{name}_score"] - ref[f"{name}_score"]).max() <= tol, name
        H.flags_agree(-r[f"{name}_score"], r[f"{name}_pred"], ref[f"{name}_dec"], ref[f"{name}_pred"], tol, name)
