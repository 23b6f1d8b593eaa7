"""Augmentation oracle against the real scipy.ndimage.affine_transform (the library Keras's
apply_affine_transform calls), and the host mirror's parameter draw / matrix against the oracle."""
import numpy as np
import pytest
from scipy import ndimage

from cellscreen.augment import ImageDataGenerator
from oracle import augment_oracle as ao


def _wide_params(rng, k):
    return ao.get_random_transform(rng, 64, 64, rotation_range=25 if k % 2 else 2, width_shift_range=0.3 if k % 3 == 0 else 0.02,
                                   height_shift_range=0.25 if k % 4 == 0 else 0.02, zoom_range=0.3 if k % 5 == 0 else 0.02)


def test_resampling_is_scipy_bit_for_bit():
    rng = np.random.RandomState(42)
    img = np.random.default_rng(0).random((64, 64)).astype(np.float32)
    for k in range(300):
        m, o = ao.affine_matrix(_wide_params(rng, k), 64, 64)
        ref = ndimage.affine_transform(img, m, o, order=1, mode="nearest")
        assert np.array_equal(ao.affine_nearest_order1(img, m, o), ref), f"draw {k}"


def test_identity_and_flips():
    img = np.random.default_rng(1).random((64, 64)).astype(np.float32)
    p = dict(theta=0.0, tx=0.0, ty=0.0, zx=1.0, zy=1.0, flip_h=False, flip_v=False)
    assert ao.affine_matrix(p, 64, 64) is None
    assert np.array_equal(ao.apply_transform(img, p), img)
    assert np.array_equal(ao.apply_transform(img, dict(p, flip_h=True)), img[:, ::-1])
    assert np.array_equal(ao.apply_transform(img, dict(p, flip_v=True)), img[::-1, :])
    assert np.array_equal(ao.apply_transform(img, dict(p, flip_h=True, flip_v=True)), img[::-1, ::-1])


def test_draw_order_and_ranges_match_the_oracle():
    gen = ImageDataGenerator.reference()
    a, b = np.random.RandomState(7), np.random.RandomState(7)
    for _ in range(100):
        p, q = gen.get_random_transform((64, 64), a), ao.get_random_transform(b, 64, 64)
        assert p == q
        assert abs(p["theta"]) <= 2 and abs(p["tx"]) <= 1.28 and abs(p["ty"]) <= 1.28
        assert 0.98 <= p["zx"] <= 1.02 and 0.98 <= p["zy"] <= 1.02
    # seven uniform draws per image, in Keras's order: theta, tx, ty, (zx, zy), flip_h, flip_v
    a, b = np.random.RandomState(3), np.random.RandomState(3)
    p = gen.get_random_transform((64, 64), a)
    u = b.uniform(-2, 2), b.uniform(-0.02, 0.02) * 64, b.uniform(-0.02, 0.02) * 64
    z = b.uniform(0.98, 1.02, 2)
    fh, fv = b.random() < 0.5, b.random() < 0.5
    assert (p["theta"], p["tx"], p["ty"], p["zx"], p["zy"], p["flip_h"], p["flip_v"]) == (u[0], u[1], u[2], z[0], z[1], fh, fv)


def test_matrix_matches_the_oracle_and_rotates_about_the_centre():
    gen = ImageDataGenerator.reference()
    rng = np.random.RandomState(11)
    for k in range(50):
        p = _wide_params(rng, k)
        (m, o), (m2, o2) = gen.affine(p, 64, 64), ao.affine_matrix(p, 64, 64)
        assert np.array_equal(m, m2) and np.array_equal(o, o2)
    # a pure rotation leaves the centre pixel position (31.5, 31.5) fixed for center = -0.5 (Keras 3)
    m, o = gen.affine(dict(theta=30.0, tx=0, ty=0, zx=1, zy=1, flip_h=False, flip_v=False), 64, 64)
    c = np.array([31.5, 31.5])
    assert np.allclose(m @ c + o, c, atol=1e-12)
    arr = gen.pack([dict(theta=0.0, tx=0.0, ty=0.0, zx=1.0, zy=1.0, flip_h=True, flip_v=False)], 64, 64)
    assert arr[0].identity == 1 and arr[0].flip_h == 1 and arr[0].flip_v == 0


def test_statistics_of_the_reference_generator():
    """Statistical parity: flips are fair, the mean shift is zero."""
    gen = ImageDataGenerator.reference()
    rng = np.random.RandomState(42)
    ps = [gen.get_random_transform((64, 64), rng) for _ in range(4000)]
    assert abs(np.mean([p["flip_h"] for p in ps]) - 0.5) < 0.03
    assert abs(np.mean([p["flip_v"] for p in ps]) - 0.5) < 0.03
    assert abs(np.mean([p["tx"] for p in ps])) < 0.05 and abs(np.mean([p["theta"] for p in ps])) < 0.08


def test_vectorised_batch_draw_is_the_scalar_algebra():
    """ImageDataGenerator.random_transforms (one numpy pass per batch, what an epoch of 1,250 x 32 draws uses) against the
    per-image path on the SAME parameters, and the packed layout against the C ABI's struct."""
    import ctypes as C
    from cellscreen import _lib as L
    gen = ImageDataGenerator.reference()
    assert gen.AFFINE_DTYPE.itemsize == C.sizeof(L.CSAugAffine)
    for f, (name, _t) in zip(gen.AFFINE_DTYPE.names, L.CSAugAffine._fields_):
        assert f == name and gen.AFFINE_DTYPE.fields[f][1] == getattr(L.CSAugAffine, name).offset
    rng = np.random.default_rng(0)
    n = 200
    th, tx, ty = rng.uniform(-2, 2, n), rng.uniform(-.02, .02, n), rng.uniform(-.02, .02, n)
    zx, zy, u1, u2 = rng.uniform(.98, 1.02, n), rng.uniform(.98, 1.02, n), rng.uniform(0, 1, n), rng.uniform(0, 1, n)

    class Replay:                      # hands the arrays out in the order random_transforms asks for them
        def __init__(self, seq):
            self.seq = list(seq)

        def uniform(self, lo, hi, size=None):
            v = self.seq.pop(0)
            assert len(v) == size and (v >= lo).all() and (v <= hi).all()
            return v
    arr = gen.random_transforms(n, (64, 64), Replay([th, tx, ty, zx, zy, u1, u2]))
    for i in range(n):
        p = dict(theta=th[i], tx=tx[i] * 64, ty=ty[i] * 64, zx=zx[i], zy=zy[i], flip_h=u1[i] < .5, flip_v=u2[i] < .5)
        m, off = gen.affine(p, 64, 64)
        assert np.abs(arr["m"][i].reshape(2, 2) - m).max() <= 1e-13 and np.abs(arr["off"][i] - off).max() <= 1e-12
        assert arr["flip_h"][i] == int(p["flip_h"]) and arr["flip_v"][i] == int(p["flip_v"]) and arr["identity"][i] == 0
    # no rotation / shift / zoom configured: every draw is the identity (+ flips)
    flips = ImageDataGenerator(horizontal_flip=True).random_transforms(50, (64, 64), np.random.default_rng(1))
    assert flips["identity"].all() and 5 < flips["flip_h"].sum() < 45 and not flips["flip_v"].any()
