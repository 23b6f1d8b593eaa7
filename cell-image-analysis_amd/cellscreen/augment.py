"""Training augmentation on the GPU: the host-side half of the reference's

    datagen = ImageDataGenerator(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02,
                                 zoom_range=0.02, horizontal_flip=True, vertical_flip=True, fill_mode='nearest')

(CAE_improved_modeltrain.py:246-254), consumed as `datagen.flow(X_train, X_train, batch_size=32)` (:287):
only the input batch is transformed, the target stays the original image.

The class keeps Keras's names and draw order (`get_random_transform`: theta, tx (height), ty (width),
zoom (zx, zy), flip_h, flip_v) and reduces each draw, in float64 and with Keras's matrix algebra
(`apply_affine_transform` + `transform_matrix_offset_center`), to the 2x2 matrix + offset that
scipy.ndimage.affine_transform takes.  The resampling itself (order=1, mode='nearest') and the flips
run in libcellscreen (`cs_train_augment`, csrc/train.hip: augment_kernel); there is no CPU fallback.

`center`: Keras 3 (the version that writes the reference's `.keras` files) centres the transform at
size/2 - 0.5; keras-preprocessing <= 1.1.0 used size/2 + 0.5.  Neither is pinned by the reference."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib as L


class ImageDataGenerator:
    def __init__(self, rotation_range=0.0, width_shift_range=0.0, height_shift_range=0.0, zoom_range=0.0,
                 horizontal_flip=False, vertical_flip=False, fill_mode="nearest", center: float = -0.5):
        if fill_mode != "nearest":
            raise NotImplementedError("the kernel implements fill_mode='nearest' (what the reference uses)")
        if np.ndim(zoom_range) == 0:
            zoom_range = (1.0 - zoom_range, 1.0 + zoom_range)
        self.rotation_range = float(rotation_range)
        self.width_shift_range = float(width_shift_range)
        self.height_shift_range = float(height_shift_range)
        self.zoom_range = (float(zoom_range[0]), float(zoom_range[1]))
        self.horizontal_flip, self.vertical_flip = bool(horizontal_flip), bool(vertical_flip)
        self.center = float(center)

    @classmethod
    def reference(cls) -> "ImageDataGenerator":
        """The generator of CAE_improved_modeltrain.py:246-254."""
        return cls(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02, zoom_range=0.02,
                   horizontal_flip=True, vertical_flip=True, fill_mode="nearest")

    # ---- Keras's parameter draw -------------------------------------------------------------
    def get_random_transform(self, img_shape, rng=np.random) -> dict:
        h, w = img_shape[0], img_shape[1]
        theta = rng.uniform(-self.rotation_range, self.rotation_range) if self.rotation_range else 0.0
        if self.height_shift_range:
            tx = rng.uniform(-self.height_shift_range, self.height_shift_range)
            if self.height_shift_range < 1:
                tx *= h
        else:
            tx = 0.0
        if self.width_shift_range:
            ty = rng.uniform(-self.width_shift_range, self.width_shift_range)
            if self.width_shift_range < 1:
                ty *= w
        else:
            ty = 0.0
        if self.zoom_range[0] == 1 and self.zoom_range[1] == 1:
            zx = zy = 1.0
        else:
            zx, zy = rng.uniform(self.zoom_range[0], self.zoom_range[1], 2)
        flip_h = bool(rng.random() < 0.5) and self.horizontal_flip
        flip_v = bool(rng.random() < 0.5) and self.vertical_flip
        return dict(theta=float(theta), tx=float(tx), ty=float(ty), zx=float(zx), zy=float(zy),
                    flip_h=bool(flip_h), flip_v=bool(flip_v))

    # ---- Keras's matrix ---------------------------------------------------------------------------
    def affine(self, p: dict, h: int, w: int):
        """(2x2 matrix, offset) of apply_affine_transform, or None for the identity."""
        m = None
        if p["theta"] != 0:
            t = np.deg2rad(p["theta"])
            m = np.array([[np.cos(t), -np.sin(t), 0.0], [np.sin(t), np.cos(t), 0.0], [0.0, 0.0, 1.0]])
        if p["tx"] != 0 or p["ty"] != 0:
            s = np.array([[1.0, 0.0, p["tx"]], [0.0, 1.0, p["ty"]], [0.0, 0.0, 1.0]])
            m = s if m is None else np.dot(m, s)
        if p["zx"] != 1 or p["zy"] != 1:
            z = np.array([[p["zx"], 0.0, 0.0], [0.0, p["zy"], 0.0], [0.0, 0.0, 1.0]])
            m = z if m is None else np.dot(m, z)
        if m is None:
            return None
        ox, oy = float(h) / 2 + self.center, float(w) / 2 + self.center
        off = np.array([[1.0, 0.0, ox], [0.0, 1.0, oy], [0.0, 0.0, 1.0]])
        rst = np.array([[1.0, 0.0, -ox], [0.0, 1.0, -oy], [0.0, 0.0, 1.0]])
        m = np.dot(np.dot(off, m), rst)
        return m[:2, :2], m[:2, 2]

    def pack(self, params: Sequence[dict], h: int, w: int):
        """The C ABI's array of cs_aug_affine for a batch of parameter dicts."""
        arr = (L.CSAugAffine * len(params))()
        for a, p in zip(arr, params):
            am = self.affine(p, h, w)
            if am is None:
                a.identity = 1
            else:
                a.identity = 0
                a.m[0], a.m[1], a.m[2], a.m[3] = am[0][0, 0], am[0][0, 1], am[0][1, 0], am[0][1, 1]
                a.off[0], a.off[1] = am[1][0], am[1][1]
            a.flip_h, a.flip_v = int(p["flip_h"]), int(p["flip_v"])
        return arr

    # ---- a whole batch at once ----------------------------------------------------------------------
    AFFINE_DTYPE = np.dtype([("m", np.float64, 4), ("off", np.float64, 2), ("identity", np.int32), ("flip_h", np.int32),
                             ("flip_v", np.int32), ("reserved", np.int32)])

    def random_transforms(self, n: int, img_shape, rng) -> np.ndarray:
        """n independent draws as one packed array of cs_aug_affine (what Trainer.augment takes): the same distributions and the
        same matrix algebra as get_random_transform + affine, evaluated with numpy over the batch -- a fit() epoch draws
        1,250 x 32 transforms and the per-image Python path costs more host time than the training step costs the GPU.  The
        draws come in a different ORDER from the stream than Keras's per-image loop takes them (all thetas first, ...), which
        only matters to someone replaying a particular seed."""
        h, w = float(img_shape[0]), float(img_shape[1])
        uni = lambda lo, hi: rng.uniform(lo, hi, n)                                  # noqa: E731
        theta = np.deg2rad(uni(-self.rotation_range, self.rotation_range)) if self.rotation_range else np.zeros(n)
        tx = uni(-self.height_shift_range, self.height_shift_range) * (h if self.height_shift_range < 1 else 1.0) if self.height_shift_range else np.zeros(n)
        ty = uni(-self.width_shift_range, self.width_shift_range) * (w if self.width_shift_range < 1 else 1.0) if self.width_shift_range else np.zeros(n)
        if self.zoom_range[0] == 1 and self.zoom_range[1] == 1:
            zx = zy = np.ones(n)
        else:
            zx, zy = uni(self.zoom_range[0], self.zoom_range[1]), uni(self.zoom_range[0], self.zoom_range[1])
        fh = (rng.uniform(0.0, 1.0, n) < 0.5) & self.horizontal_flip
        fv = (rng.uniform(0.0, 1.0, n) < 0.5) & self.vertical_flip
        c, s_ = np.cos(theta), np.sin(theta)
        # (rotation . shift . zoom): linear part R Z, translation R t; then moved to the image centre: off = o + R t - (R Z) o
        m00, m01, m10, m11 = c * zx, -s_ * zy, s_ * zx, c * zy
        ox, oy = h / 2 + self.center, w / 2 + self.center
        out = np.zeros(n, self.AFFINE_DTYPE)
        out["m"][:, 0], out["m"][:, 1], out["m"][:, 2], out["m"][:, 3] = m00, m01, m10, m11
        out["off"][:, 0] = ox + (c * tx - s_ * ty) - (m00 * ox + m01 * oy)
        out["off"][:, 1] = oy + (s_ * tx + c * ty) - (m10 * ox + m11 * oy)
        out["identity"] = (theta == 0) & (tx == 0) & (ty == 0) & (zx == 1) & (zy == 1)
        out["flip_h"], out["flip_v"] = fh, fv
        return out

    # ---- the keyed draws of cs_train_fit_step --------------------------------------------------------
    def config(self) -> "L.CSAugConfig":
        """This generator as the C ABI's cs_aug_config (cs_train_fit_step draws the transforms itself)."""
        c = L.CSAugConfig()
        c.rotation_range = self.rotation_range
        c.width_shift_range, c.height_shift_range = self.width_shift_range, self.height_shift_range
        c.zoom_lo, c.zoom_hi = self.zoom_range
        c.horizontal_flip, c.vertical_flip = int(self.horizontal_flip), int(self.vertical_flip)
        c.center = self.center
        return c

    def keyed_transforms(self, seed: int, step: int, n: int, img_shape) -> np.ndarray:
        """The n transforms cs_train_fit_step draws for fit() step `step` (csrc/train_api.hip: fit_transform): image b's seven
        uniforms are counter_uniforms(seed, step, n)[b] in Keras's get_random_transform order, reduced with random_transforms'
        closed form.  Host mirror of the C function: tests hold the two together."""
        h, w = float(img_shape[0]), float(img_shape[1])
        u = counter_uniforms(seed, step, n)
        lo_hi = lambda r, col: -r + (r - -r) * u[:, col]                               # noqa: E731  numpy's uniform(lo, hi): lo + (hi - lo) u
        theta = (lo_hi(self.rotation_range, 0) if self.rotation_range else np.zeros(n)) * (np.pi / 180.0)
        tx = lo_hi(self.height_shift_range, 1) * (h if self.height_shift_range < 1 else 1.0) if self.height_shift_range else np.zeros(n)
        ty = lo_hi(self.width_shift_range, 2) * (w if self.width_shift_range < 1 else 1.0) if self.width_shift_range else np.zeros(n)
        if self.zoom_range[0] == 1 and self.zoom_range[1] == 1:
            zx = zy = np.ones(n)
        else:
            zx = self.zoom_range[0] + (self.zoom_range[1] - self.zoom_range[0]) * u[:, 3]
            zy = self.zoom_range[0] + (self.zoom_range[1] - self.zoom_range[0]) * u[:, 4]
        c, s_ = np.cos(theta), np.sin(theta)
        m00, m01, m10, m11 = c * zx, -s_ * zy, s_ * zx, c * zy
        ox, oy = h / 2 + self.center, w / 2 + self.center
        out = np.zeros(n, self.AFFINE_DTYPE)
        out["m"][:, 0], out["m"][:, 1], out["m"][:, 2], out["m"][:, 3] = m00, m01, m10, m11
        out["off"][:, 0] = ox + (c * tx - s_ * ty) - (m00 * ox + m01 * oy)
        out["off"][:, 1] = oy + (s_ * tx + c * ty) - (m10 * ox + m11 * oy)
        out["identity"] = (theta == 0) & (tx == 0) & (ty == 0) & (zx == 1) & (zy == 1)
        out["flip_h"] = (u[:, 5] < 0.5) & self.horizontal_flip
        out["flip_v"] = (u[:, 6] < 0.5) & self.vertical_flip
        return out

    # ---- flow ---------------------------------------------------------------------------------
    def random_batch(self, trainer, batch, rng=np.random):
        """What one `next(datagen.flow(x, ...))` does to x: one independent draw per image.
        Returns (augmented batch, list of the drawn parameter dicts)."""
        n, h, w = batch.shape[0], batch.shape[1], batch.shape[2]
        params: List[dict] = [self.get_random_transform((h, w), rng) for _ in range(n)]
        return trainer.augment(batch, self.pack(params, h, w)), params


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64's finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def counter_uniforms(seed: int, step: int, n: int) -> np.ndarray:
    """u[b, j] in [0, 1), b < n, j < 7: the counter-based draws of cs_train_fit_step (csrc/train_api.hip: fit_u01) -- three
    rounds of splitmix64's finaliser over (seed, step, 8 b + j), top 53 bits.  A step's draws depend on its key alone."""
    key = _mix64(_mix64(np.array([seed], dtype=np.uint64)) ^ np.uint64(step))
    ctr = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(7, dtype=np.uint64)[None, :])
    h = _mix64(key ^ ctr)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def draw_transforms_c(gen: "ImageDataGenerator", seed: int, step: int, n: int, img_shape) -> np.ndarray:
    """cs_train_draw_transforms itself (host-only entry of the library), as an AFFINE_DTYPE array."""
    out = np.zeros(n, ImageDataGenerator.AFFINE_DTYPE)
    cfg = gen.config()
    L.check(L.load_library().cs_train_draw_transforms(C.byref(cfg), int(seed), int(step), n, int(img_shape[0]), int(img_shape[1]),
                                                      out.ctypes.data))
    return out


class _GeneratorRng:
    """Adapts numpy's Generator (used by cellscreen.training for shuffling) to the .uniform/.random
    interface of the legacy np.random module that Keras draws from."""

    def __init__(self, g):
        self._g = g

    def uniform(self, lo, hi, size=None):
        return self._g.uniform(lo, hi, size)

    def random(self):
        return self._g.random()


def reference_augment(trainer):
    """augment(batch, rng) hook for ImprovedAnomalyDetectionTraining: the reference's generator."""
    gen = ImageDataGenerator.reference()

    def hook(batch, rng):
        r = _GeneratorRng(rng) if isinstance(rng, np.random.Generator) else rng
        return gen.random_batch(trainer, batch, r)[0]
    return hook
