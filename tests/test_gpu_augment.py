"""cs_train_augment (HIP) against scipy.ndimage.affine_transform + flips on the same transforms.
Coordinates and interpolation are fp64 on both sides, the result is rounded to float32: tolerance one
float32 rounding (6e-8 on [0,1) data); the test also reports how many pixels are bit-identical."""
import numpy as np
import pytest
from scipy import ndimage

from cellscreen import synth
from cellscreen.augment import ImageDataGenerator, reference_augment
from cellscreen.trainer import Trainer
from oracle import augment_oracle as ao

pytestmark = pytest.mark.gpu

TOL = 6e-8


@pytest.fixture(scope="module")
def trainer():
    t = Trainer(synth.random_cae(seed=42, trivial_bn=True))
    yield t
    t.close()


def _scipy_apply(gen, img, p):
    am = gen.affine(p, 64, 64)
    out = img if am is None else ndimage.affine_transform(img, am[0], am[1], order=1, mode="nearest")
    if p["flip_h"]:
        out = out[:, ::-1]
    if p["flip_v"]:
        out = out[::-1, :]
    return out


def test_reference_generator_matches_scipy(trainer):
    gen = ImageDataGenerator.reference()
    x = synth.blob_crops(3, 96)
    rng = np.random.RandomState(42)
    got, params = gen.random_batch(trainer, x, rng)
    want = np.stack([_scipy_apply(gen, im, p) for im, p in zip(x, params)])
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert err.max() <= TOL, f"max err {err.max():.3e}"
    assert (got == want).mean() > 0.99
    assert not np.array_equal(got, x)                     # something was actually transformed


def test_large_transforms_edges_and_identity(trainer):
    gen = ImageDataGenerator(rotation_range=40, width_shift_range=0.4, height_shift_range=0.4, zoom_range=0.5,
                             horizontal_flip=True, vertical_flip=True)
    x = synth.synth_crops(5, 0, 64)                        # white noise: the worst case for interpolation parity
    rng = np.random.RandomState(1)
    params = [gen.get_random_transform((64, 64), rng) for _ in range(len(x))]
    params[0] = dict(theta=0.0, tx=0.0, ty=0.0, zx=1.0, zy=1.0, flip_h=False, flip_v=False)      # identity
    params[1] = dict(theta=0.0, tx=0.0, ty=0.0, zx=1.0, zy=1.0, flip_h=True, flip_v=True)        # flips only
    params[2] = dict(theta=0.0, tx=100.0, ty=-100.0, zx=1.0, zy=1.0, flip_h=False, flip_v=False)  # wholly outside: edge fill
    got = trainer.augment(x, gen.pack(params, 64, 64))
    assert np.array_equal(got[0], x[0]) and np.array_equal(got[1], x[1][::-1, ::-1])
    for k, (im, p) in enumerate(zip(x, params)):
        assert np.abs(got[k].astype(np.float64) - _scipy_apply(gen, im, p)).max() <= TOL, f"image {k}: {p}"
        assert np.abs(got[k].astype(np.float64) - ao.apply_transform(im, p)).max() <= TOL


def test_device_tensors_and_arguments(trainer):
    import torch
    gen = ImageDataGenerator.reference()
    x = synth.blob_crops(9, 32)
    rng = np.random.RandomState(5)
    params = [gen.get_random_transform((64, 64), rng) for _ in range(32)]
    tf = gen.pack(params, 64, 64)
    host = trainer.augment(x, tf)
    dev = trainer.augment(torch.from_numpy(x).cuda(), tf)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), host)
    assert trainer.augment(x[:0], gen.pack([], 64, 64)).shape == (0, 64, 64)


def test_training_hook_augments_input_only(trainer):
    """datagen.flow(X_train, X_train): the step sees an augmented input and the original target."""
    hook = reference_augment(trainer)
    yb = synth.blob_crops(4, 32)
    xb = hook(yb, np.random.default_rng(42))
    assert xb.shape == yb.shape and xb.dtype == np.float32 and not np.array_equal(xb, yb)
    loss, mae = trainer.forward_backward(xb, yb)
    assert np.isfinite(loss) and np.isfinite(mae)
