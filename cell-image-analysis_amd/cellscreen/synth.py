"""Synthetic inputs and random-init weights (there are no model files, images or fixtures
anywhere in the reference: its .gitignore:105-129 excludes them all).

* synth_crops: the counter-based U[0,1) generator, bit-identical to the HIP kernel
  (csrc/detector.hip:hash24) and to oracle/cae_oracle.c:orc_hash24.
* blob_crops: structured crops (1-3 Gaussian blobs + noise) for training runs, so the
  reconstruction loss means something.
* random_cae: Glorot-uniform kernels and zero biases (Keras Conv2D defaults) with
  non-trivial BatchNormalization statistics so BN is exercised.
"""
from __future__ import annotations

import numpy as np

from . import spec
from .spec import CAEWeights

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def synth_crops(seed: int, first_cell: int, n: int, hw=spec.INPUT_HW) -> np.ndarray:
    """(n, H, W) float32, value = hash24(seed, cell, pixel) / 2^24."""
    npix = hw[0] * hw[1]
    with np.errstate(over="ignore"):
        cell = (np.arange(n, dtype=np.uint64) + np.uint64(first_cell))[:, None]
        pix = np.arange(npix, dtype=np.uint64)[None, :]
        z = np.uint64(seed) + cell * np.uint64(0x9E3779B97F4A7C15) + pix * np.uint64(0xD1B54A32D192ED03)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        h = (z >> np.uint64(40)).astype(np.uint32)
    return (h.astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(n, hw[0], hw[1])


def blob_crops(seed: int, n: int, hw=spec.INPUT_HW) -> np.ndarray:
    """(n, H, W) float32 in [0,1]: sum of 1-3 Gaussian blobs + 0.05 noise, clipped."""
    rng = np.random.default_rng(seed)
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.empty((n, H, W), dtype=np.float32)
    for i in range(n):
        img = np.zeros((H, W), dtype=np.float32)
        for _ in range(rng.integers(1, 4)):
            cy, cx = rng.uniform(0.25 * H, 0.75 * H), rng.uniform(0.25 * W, 0.75 * W)
            sy, sx = rng.uniform(0.08 * H, 0.22 * H), rng.uniform(0.08 * W, 0.22 * W)
            amp = rng.uniform(0.4, 1.0)
            img += amp * np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2)).astype(np.float32)
        img += rng.normal(0.0, 0.05, size=(H, W)).astype(np.float32)
        out[i] = np.clip(img, 0.0, 1.0)
    return out


def raw_crops(seed: int, n: int, dtype=np.uint8, min_side: int = 8, max_side: int = 120, flat_every: int = 0):
    """n ragged bounding-box crops as a microscope channel delivers them (uint8 / uint16): a bright
    cell-like blob with spots on a dim noisy background, sides uniform in [min_side, max_side].
    With flat_every = k every k-th crop is low-contrast (CLAHE's clip/redistribute path works hard)."""
    rng = np.random.default_rng(seed)
    top = 255 if np.dtype(dtype) == np.uint8 else 65535
    out = []
    for i in range(n):
        H, W = int(rng.integers(min_side, max_side + 1)), int(rng.integers(min_side, max_side + 1))
        yy, xx = np.mgrid[0:H, 0:W]
        cy, cx = rng.uniform(0.35 * H, 0.65 * H), rng.uniform(0.35 * W, 0.65 * W)
        sy, sx = rng.uniform(0.15 * H, 0.3 * H), rng.uniform(0.15 * W, 0.3 * W)
        img = 0.08 + 0.7 * np.exp(-(((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        for _ in range(3):
            py, px = rng.integers(0, H), rng.integers(0, W)
            img += 0.3 * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / rng.uniform(2.0, 6.0))
        img += rng.normal(0.0, 0.04, size=(H, W))
        if flat_every and i % flat_every == flat_every - 1:
            img = 0.5 + 0.01 * img
        out.append(np.round(np.clip(img, 0.0, 1.0) * top).astype(dtype))
    return out


def random_cae(seed: int = 42, hw=spec.INPUT_HW, channels=spec.CHANNELS, n_enc=spec.N_ENC,
               trivial_bn: bool = False) -> CAEWeights:
    """Glorot-uniform convs, zero biases (Keras defaults); BN gamma in [0.5,1.5], beta in
    [-0.1,0.1], moving mean in [0,0.5], moving var in [0.5,1.5] unless trivial_bn (Keras init:
    gamma=1, beta=0, mean=0, var=1)."""
    rng = np.random.default_rng(seed)
    ks, bs, g, b, m, v = [], [], [], [], [], []
    cin = 1
    for l, cout in enumerate(channels):
        limit = np.sqrt(6.0 / (9 * cin + 9 * cout))
        ks.append(rng.uniform(-limit, limit, size=(3, 3, cin, cout)).astype(np.float32))
        bs.append(np.zeros(cout, dtype=np.float32))
        if l < len(channels) - 1:
            if trivial_bn:
                g.append(np.ones(cout, np.float32)); b.append(np.zeros(cout, np.float32))
                m.append(np.zeros(cout, np.float32)); v.append(np.ones(cout, np.float32))
            else:
                g.append(rng.uniform(0.5, 1.5, cout).astype(np.float32))
                b.append(rng.uniform(-0.1, 0.1, cout).astype(np.float32))
                m.append(rng.uniform(0.0, 0.5, cout).astype(np.float32))
                v.append(rng.uniform(0.5, 1.5, cout).astype(np.float32))
        cin = cout
    return CAEWeights(ks, bs, g, b, m, v, tuple(hw), n_enc, spec.BN_EPS).validate()


def perturbed_encoder(ae: CAEWeights, seed: int = 7, rel: float = 1e-2) -> CAEWeights:
    """An encoder.keras weight set that differs from the autoencoder's encoder half, as happens
    when ModelCheckpoint's best epoch is not the final one (CAE...:270-275 vs :300)."""
    rng = np.random.default_rng(seed)
    e = ae.encoder_half()
    jit = lambda a: (a * (1.0 + rel * rng.standard_normal(a.shape))).astype(np.float32)
    return CAEWeights([jit(k) for k in e.kernels], [b + np.float32(rel) * rng.standard_normal(b.shape).astype(np.float32) for b in e.biases],
                      [jit(x) for x in e.bn_gamma], [jit(x) for x in e.bn_beta], [jit(x) for x in e.bn_mean],
                      [jit(x) for x in e.bn_var], e.input_hw, e.n_enc, e.bn_eps).validate()
