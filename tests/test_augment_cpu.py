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


def test_keyed_transforms_of_fit_step_match_the_library_and_keras_draw():
    """cs_train_fit_step draws a batch's transforms itself from a counter-based generator keyed (seed, step, image): the C function
    (cs_train_draw_transforms, host only) against its Python mirror -- the same seven uniforms per image in Keras's
    get_random_transform order (theta, tx = height shift, ty = width shift, zx, zy, flip_h, flip_v), reduced with the closed form of
    random_transforms -- and that closed form against the per-image Keras matrix algebra (affine())."""
    from cellscreen import augment as A
    gens = [A.ImageDataGenerator.reference(),
            A.ImageDataGenerator(rotation_range=25, width_shift_range=0.3, height_shift_range=3.0, zoom_range=(0.7, 1.4), horizontal_flip=True),
            A.ImageDataGenerator(zoom_range=0.0, vertical_flip=True),                      # nothing but a flip: identity resampling
            A.ImageDataGenerator(rotation_range=10, center=0.5)]
    for gi, g in enumerate(gens):
        for seed, step, shape in ((42, 0, (64, 64)), (43, 1249, (64, 64)), (2 ** 40 + 7, 10 ** 9, (128, 96))):
            py = g.keyed_transforms(seed, step, 32, shape)
            c = A.draw_transforms_c(g, seed, step, 32, shape)
            for k in ("m", "off"):
                assert np.allclose(py[k], c[k], rtol=0, atol=1e-12), (gi, k)                # the same doubles up to an fma contraction
            for k in ("identity", "flip_h", "flip_v"):
                assert np.array_equal(py[k], c[k]), (gi, k)
            # ... and the closed form is Keras's matrix product for the same parameters
            u = A.counter_uniforms(seed, step, 32)
            for b in (0, 5, 31):
                r = g.rotation_range
                p = dict(theta=(-r + 2 * r * u[b, 0]) if r else 0.0,
                         tx=(-g.height_shift_range + 2 * g.height_shift_range * u[b, 1]) * (shape[0] if g.height_shift_range < 1 else 1.0) if g.height_shift_range else 0.0,
                         ty=(-g.width_shift_range + 2 * g.width_shift_range * u[b, 2]) * (shape[1] if g.width_shift_range < 1 else 1.0) if g.width_shift_range else 0.0,
                         zx=g.zoom_range[0] + (g.zoom_range[1] - g.zoom_range[0]) * u[b, 3] if g.zoom_range != (1.0, 1.0) else 1.0,
                         zy=g.zoom_range[0] + (g.zoom_range[1] - g.zoom_range[0]) * u[b, 4] if g.zoom_range != (1.0, 1.0) else 1.0,
                         flip_h=bool(u[b, 5] < 0.5) and g.horizontal_flip, flip_v=bool(u[b, 6] < 0.5) and g.vertical_flip)
                am = g.affine(p, shape[0], shape[1])
                if am is None:
                    assert c["identity"][b] == 1
                else:
                    assert np.allclose(np.ravel(am[0]), c["m"][b], rtol=0, atol=1e-12) and np.allclose(am[1], c["off"][b], rtol=0, atol=1e-10)
                assert bool(c["flip_h"][b]) == p["flip_h"] and bool(c["flip_v"][b]) == p["flip_v"]
    # the generator: uniform, and a step's draws depend on its key alone
    u = np.concatenate([A.counter_uniforms(7, s, 64) for s in range(200)])
    assert u.shape == (12800, 7) and 0.0 <= u.min() and u.max() < 1.0
    assert np.abs(u.mean(axis=0) - 0.5).max() < 0.02 and np.abs(u.var(axis=0) - 1.0 / 12).max() < 0.01
    assert np.abs(np.corrcoef(u.T) - np.eye(7)).max() < 0.05
    assert np.array_equal(A.counter_uniforms(7, 3, 64)[:32], A.counter_uniforms(7, 3, 32))
    assert not np.array_equal(A.counter_uniforms(7, 3, 32), A.counter_uniforms(7, 4, 32))
    assert not np.array_equal(A.counter_uniforms(7, 3, 32), A.counter_uniforms(8, 3, 32))
