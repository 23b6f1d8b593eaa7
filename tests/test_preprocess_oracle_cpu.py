"""The preprocess oracle against the real scikit-image / SciPy outputs committed in
tests/golden/golden_preprocess.npz (made by tests/golden/make_golden_preprocess.py under the conda
interpreter: scikit-image 0.18.3, SciPy 1.7.1), plus the host-side packing of the C ABI."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cellscreen import preprocess as pp
from cellscreen import synth
from oracle import preprocess_oracle as po


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "golden_preprocess.npz"))


def test_golden_was_made_by_the_real_libraries(gold):
    v = [str(x) for x in gold["versions"]]
    assert v[0] == "scikit-image 0.18.3" and v[1] == "scipy 1.7.1"


def test_clahe_stage_is_bit_exact(gold):
    for i in range(int(gold["n"])):
        u = po.clahe_u16(gold[f"crop_{i}"], float(gold["clip_limit"]))
        assert u.dtype == np.uint16
        assert np.array_equal(u, gold[f"clahe_u16_{i}"]), f"crop {i}"


def test_equalize_adapthist_is_exact(gold):
    for i in range(int(gold["n"])):
        eq = po.equalize_adapthist(gold[f"crop_{i}"], float(gold["clip_limit"]))
        assert np.array_equal(eq, gold[f"eq_{i}"]), f"crop {i}"       # same float64 bits


def test_resize_matches_to_1e12(gold):
    for i in range(int(gold["n"])):
        out = po.resize_to_64(gold[f"eq_{i}"])
        assert out.shape == (64, 64)
        assert np.abs(out - gold[f"out_{i}"]).max() <= 1e-12, f"crop {i}"


def test_full_chain_matches(gold):
    crops = [gold[f"crop_{i}"] for i in range(int(gold["n"])) if gold[f"crop_{i}"].dtype == np.uint8]
    want = [gold[f"out_{i}"] for i in range(int(gold["n"])) if gold[f"crop_{i}"].dtype == np.uint8]
    got = po.preprocess_crops(crops)
    assert got.dtype == np.float32 and got.shape == (len(crops), 64, 64)
    for g, w in zip(got, want):
        assert np.abs(g.astype(np.float64) - w).max() <= 6e-8          # one float32 rounding


def test_resize_is_identity_at_64(gold):
    """at 64x64 the resize does nothing (sigma = 0, integer sample points)."""
    eq = gold["eq_6"]
    assert eq.shape == (64, 64)
    assert np.abs(po.resize_to_64(eq) - eq).max() <= 1e-13


def test_small_crop_is_rejected():
    with pytest.raises(ValueError):
        po.clahe_u16(np.zeros((7, 20), np.uint8))


def test_clip_histogram_invariants():
    rng = np.random.default_rng(3)
    for _ in range(50):
        npix = int(rng.integers(16, 900))
        h = np.bincount(rng.integers(0, int(rng.integers(2, 256)), npix), minlength=256)
        clim = int(max(0.02 * npix, 1))
        c = po.clip_histogram(h, clim)
        assert c.min() >= 0 and c.max() <= clim                  # redistribution never lifts a bin past the limit
        assert (c[h == 0] <= clim).all() and c.sum() <= 256 * clim


def test_pack_crops_layout():
    crops = synth.raw_crops(5, 7, np.uint16, 8, 30)
    pix, off, hs, ws = pp.pack_crops(crops)
    assert pix.dtype == np.uint16 and off.dtype == np.int64 and hs.dtype == np.int32
    assert off[0] == 0 and pix.size == sum(c.size for c in crops)
    for c, o, h, w in zip(crops, off, hs, ws):
        assert np.array_equal(pix[o:o + h * w].reshape(h, w), c)
    with pytest.raises(TypeError):
        pp.pack_crops([np.zeros((8, 8), np.float32)])
    with pytest.raises(TypeError):
        pp.pack_crops([np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint16)])
    e = pp.pack_crops([])
    assert e[0].size == 0 and e[1].size == 0
