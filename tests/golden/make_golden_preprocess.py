"""Generates tests/golden/golden_preprocess.npz with the REAL libraries the reference calls.

Run with the image's conda interpreter (scikit-image 0.18.3, SciPy 1.7.1, numpy 1.26.4):

    /opt/conda/bin/python3.9 tests/golden/make_golden_preprocess.py

For a set of synthetic bounding-box crops (uint8 / uint16, ragged sizes) it stores the input and
what improved_detection.py:98-99 computes from it:
    clahe_u16_i : the uint16 image skimage's `_clahe` returns inside equalize_adapthist
    eq_i        : exposure.equalize_adapthist(crop, clip_limit=0.02)          (float64)
    out_i       : resize(eq_i, (64, 64), anti_aliasing=True)                  (float64)
Nothing from /root/reference is imported: the two calls are library calls.
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage import exposure, img_as_uint                      # noqa: E402
from skimage.exposure import _adapthist, rescale_intensity     # noqa: E402
from skimage.transform import resize                           # noqa: E402

CLIP = 0.02

# (H, W, dtype, kind)
CASES = [
    (8, 8, "u8", "noise"), (9, 15, "u8", "blob"), (16, 16, "u16", "blob"), (23, 31, "u8", "blob"),
    (37, 52, "u16", "blob"), (48, 40, "u8", "blob"), (64, 64, "u8", "blob"), (64, 64, "u16", "noise"),
    (65, 63, "u16", "blob"), (80, 96, "u8", "blob"), (100, 71, "u16", "blob"), (127, 128, "u8", "blob"),
    (181, 97, "u16", "blob"), (40, 200, "u8", "blob"), (33, 33, "u8", "const"), (56, 72, "u16", "flat"),
    (72, 56, "u8", "sat"), (15, 120, "u16", "noise"),
]


def make_crop(rng, H, W, dt, kind):
    top = 255 if dt == "u8" else 65535
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == "const":
        img = np.full((H, W), 0.37)
    elif kind == "noise":
        img = rng.random((H, W))
    else:
        cy, cx = H * (0.35 + 0.3 * rng.random()), W * (0.35 + 0.3 * rng.random())
        sy, sx = H * (0.15 + 0.15 * rng.random()), W * (0.15 + 0.15 * rng.random())
        img = 0.08 + 0.7 * np.exp(-(((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        for _ in range(3):                                      # bright organelle-like spots
            py, px = rng.integers(0, H), rng.integers(0, W)
            img += 0.3 * np.exp(-(((yy - py) ** 2 + (xx - px) ** 2) / (2.0 + 4 * rng.random())))
        img += 0.04 * rng.standard_normal((H, W))
        if kind == "flat":
            img = 0.5 + 0.01 * img                              # nearly constant: heavy clipping
        if kind == "sat":
            img = img * 2.5                                     # saturates at the dtype maximum
    img = np.clip(img, 0, 1)
    return np.round(img * top).astype(np.uint8 if dt == "u8" else np.uint16)


def clahe_u16(image):
    """The internal uint16 stage, called exactly as equalize_adapthist does (_adapthist.py:78-92)."""
    im = img_as_uint(image)
    im = np.round(rescale_intensity(im, out_range=(0, _adapthist.NR_OF_GRAY - 1))).astype(np.uint16)
    ks = [int(im.shape[d] // 8) for d in range(im.ndim)]
    return _adapthist._clahe(im, ks, CLIP, 256)


def main():
    rng = np.random.default_rng(20240607)
    out = {"n": np.int64(len(CASES)), "clip_limit": np.float64(CLIP)}
    for i, (H, W, dt, kind) in enumerate(CASES):
        crop = make_crop(rng, H, W, dt, kind)
        eq = exposure.equalize_adapthist(crop, clip_limit=CLIP)
        rs = resize(eq, (64, 64), anti_aliasing=True)
        out[f"crop_{i}"] = crop
        out[f"clahe_u16_{i}"] = clahe_u16(crop)
        out[f"eq_{i}"] = eq
        out[f"out_{i}"] = rs
    import skimage, scipy
    out["versions"] = np.array([f"scikit-image {skimage.__version__}", f"scipy {scipy.__version__}",
                                f"numpy {np.__version__}"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
