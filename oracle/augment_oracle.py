"""CPU oracle of the training augmentation -- TEST INFRASTRUCTURE ONLY (see cae_oracle.c header).

Restates what `ImageDataGenerator(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02,
zoom_range=0.02, horizontal_flip=True, vertical_flip=True, fill_mode='nearest')` does to one image
(CAE_improved_modeltrain.py:246-254, applied to the INPUT batch only by `datagen.flow(X_train, X_train)`
at :287): draw (theta, tx, ty, zx, zy, flip_h, flip_v) from numpy's global RNG in Keras's order,
build the 3x3 affine matrix about the image centre, resample with
`scipy.ndimage.affine_transform(order=1, mode='nearest')`, then flip.

Two halves with different pinning:
  * the resampling is SciPy's -- pinned: tests compare `affine_nearest_order1` (explicit numpy bilinear)
    with the real `scipy.ndimage.affine_transform` of the installed SciPy;
  * the matrix construction and the draw order are Keras's (`keras/src/legacy/preprocessing/image.py`:
    `get_random_transform`, `apply_affine_transform`, `transform_matrix_offset_center`).  Keras is absent
    from this image and the reference pins no version: PARITY UNPINNED for that half, restated from the
    published source.  One constant differs between releases -- the centre is `size/2 - 0.5` in Keras 3
    (which writes the `.keras` files the reference saves) and `size/2 + 0.5` in keras-preprocessing <= 1.1.0;
    it is the `center` argument here and in the HIP kernel (default: Keras 3).
The RNG stream itself (numpy's global Mersenne Twister shared with the shuffling) is not reproduced on
the device: parameters are drawn on the host in the same order and handed to the kernel.
"""
from __future__ import annotations

import numpy as np

PARAM_FIELDS = ("theta", "tx", "ty", "zx", "zy", "flip_h", "flip_v")


def get_random_transform(rng, h, w, rotation_range=2.0, width_shift_range=0.02, height_shift_range=0.02,
                         zoom_range=0.02, horizontal_flip=True, vertical_flip=True):
    """One parameter draw in Keras's order.  `rng` needs .uniform/.random (np.random module or a
    RandomState): theta, tx (height!), ty (width), [shear skipped: 0], (zx, zy), flip_h, flip_v."""
    theta = rng.uniform(-rotation_range, rotation_range) if rotation_range else 0.0
    if height_shift_range:
        tx = rng.uniform(-height_shift_range, height_shift_range)
        if height_shift_range < 1:
            tx *= h
    else:
        tx = 0.0
    if width_shift_range:
        ty = rng.uniform(-width_shift_range, width_shift_range)
        if width_shift_range < 1:
            ty *= w
    else:
        ty = 0.0
    lo, hi = 1.0 - zoom_range, 1.0 + zoom_range            # scalar zoom_range -> [1 - z, 1 + z]
    if lo == 1.0 and hi == 1.0:
        zx = zy = 1.0
    else:
        zx, zy = rng.uniform(lo, hi, 2)
    flip_h = bool(rng.random() < 0.5) and horizontal_flip
    flip_v = bool(rng.random() < 0.5) and vertical_flip
    return dict(theta=float(theta), tx=float(tx), ty=float(ty), zx=float(zx), zy=float(zy),
                flip_h=bool(flip_h), flip_v=bool(flip_v))


def affine_matrix(p, h, w, center=-0.5):
    """apply_affine_transform's matrix: rotation . shift . zoom, conjugated by the centre offset.
    Returns (2x2 matrix, offset) as scipy.ndimage.affine_transform takes them (output -> input coords),
    or None when the transform is the identity (Keras then skips the resampling)."""
    m = None
    if p["theta"] != 0:
        t = np.deg2rad(p["theta"])
        m = np.array([[np.cos(t), -np.sin(t), 0.0], [np.sin(t), np.cos(t), 0.0], [0.0, 0.0, 1.0]])
    if p["tx"] != 0 or p["ty"] != 0:
        s = np.array([[1.0, 0.0, p["tx"]], [0.0, 1.0, p["ty"]], [0.0, 0.0, 1.0]])
        m = s if m is None else m @ s
    if p["zx"] != 1 or p["zy"] != 1:
        z = np.array([[p["zx"], 0.0, 0.0], [0.0, p["zy"], 0.0], [0.0, 0.0, 1.0]])
        m = z if m is None else m @ z
    if m is None:
        return None
    ox, oy = float(h) / 2 + center, float(w) / 2 + center
    off = np.array([[1.0, 0.0, ox], [0.0, 1.0, oy], [0.0, 0.0, 1.0]])
    rst = np.array([[1.0, 0.0, -ox], [0.0, 1.0, -oy], [0.0, 0.0, 1.0]])
    m = off @ m @ rst
    return m[:2, :2].copy(), m[:2, 2].copy()


def affine_nearest_order1(img, mat, offset):
    """scipy.ndimage.affine_transform(img, mat, offset, order=1, mode='nearest') for a 2-D image:
    bilinear interpolation at mat @ (r, c) + offset, coordinates clamped to the image (beyond the
    edge the nearest-extended image is constant), float64 arithmetic, result cast to img.dtype."""
    h, w = img.shape
    r, c = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    rr = mat[0, 0] * r + mat[0, 1] * c + offset[0]
    cc = mat[1, 0] * r + mat[1, 1] * c + offset[1]
    rr = np.clip(rr, 0.0, h - 1.0)
    cc = np.clip(cc, 0.0, w - 1.0)
    r0 = np.minimum(np.floor(rr), h - 2).astype(np.int64) if h > 1 else np.zeros_like(rr, np.int64)
    c0 = np.minimum(np.floor(cc), w - 2).astype(np.int64) if w > 1 else np.zeros_like(cc, np.int64)
    fr, fc = rr - r0, cc - c0
    x = img.astype(np.float64)
    r1, c1 = np.minimum(r0 + 1, h - 1), np.minimum(c0 + 1, w - 1)
    out = (1 - fr) * ((1 - fc) * x[r0, c0] + fc * x[r0, c1]) + fr * ((1 - fc) * x[r1, c0] + fc * x[r1, c1])
    return out.astype(img.dtype)


def apply_transform(img, p, center=-0.5):
    """ImageDataGenerator.apply_transform for one (H, W) image: affine resample, then the flips."""
    am = affine_matrix(p, img.shape[0], img.shape[1], center)
    out = img if am is None else affine_nearest_order1(img, am[0], am[1])
    if p["flip_h"]:
        out = out[:, ::-1]
    if p["flip_v"]:
        out = out[::-1, :]
    return np.ascontiguousarray(out)


def augment_batch(batch, params, center=-0.5):
    return np.stack([apply_transform(x, p, center) for x, p in zip(batch, params)])
