#!/usr/bin/env python3
"""Numerical study (CPU, numpy): could the f32 contractions of the Winograd convs run on the bf16 matrix cores as split products?

On gfx950 the f32-input MFMA runs at the f32 VECTOR rate (1/16 of the bf16 MFMA rate), and every conv kernel of the path is bound
by it.  Writing a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (and b likewise), a
product a b is a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) up to 2^-24 -- 6 bf16 products ("bf16x3"), each exact in the MFMA's f32
accumulator; 3 products (a1b1 + a1b2 + a2b1, "bf16x2") keep 16 bits.  This script measures what that does to conv2 as
F(4x4,3x3) (the contraction M[xi] = V[xi] U[xi] only; transforms stay f32) against the bars of tests/helpers.py, the same way
tests/study_wino_error.py measured F(4x4,3x3) itself before it was adopted.  It decides nothing in the product: it is the
measurement a split-bf16 kernel would have to be justified by.  Lives under tests/ because it uses the CPU oracle."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import study_wino_error as W  # noqa: E402
from cellscreen import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def bf16(x):
    """round-to-nearest-even float32 -> bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split(x, parts):
    out, rem = [], x.astype(np.float32)
    for _ in range(parts):
        p = bf16(rem)
        out.append(p)
        rem = (rem - p).astype(np.float32)
    return out


def contract(V, U, mode):
    """M[t, xi, co] = sum_k V[t, xi, k] U[xi, k, co].  mode 'f32': k-ordered fp32 fma chain; 'x3' / 'x2': split-bf16 products,
    each product exact, accumulated in fp32 in k order (largest terms first within a k, as a kernel would issue them)."""
    T, X, K = V.shape
    Co = U.shape[2]
    M = np.zeros((T, X, Co), np.float32)
    if mode == "f32":
        for k in range(K):
            M = (M.astype(np.float64) + V[:, :, k, None].astype(np.float64) * U[None, :, k, :].astype(np.float64)).astype(np.float32)
        return M
    parts = 3 if mode == "x3" else 2
    Vs, Us = split(V, parts), split(U, parts)
    pairs = [(0, 0), (0, 1), (1, 0)] + ([(0, 2), (1, 1), (2, 0)] if parts == 3 else [])
    # the matrix core accumulates K = 32 bf16 products per instruction: model each (pair, k-block of 32) as one exact dot
    # product rounded once into the fp32 accumulator (an upper bound on its accuracy; per-product rounding is studied below)
    for i, j in pairs[::-1]:                                   # small terms first
        for k0 in range(0, K, 32):
            blk = np.einsum("txk,xkc->txc", Vs[i][:, :, k0:k0 + 32].astype(np.float64), Us[j][:, k0:k0 + 32, :].astype(np.float64))
            M = (M.astype(np.float64) + blk).astype(np.float32)
    return M


def main():
    n = int(os.environ.get("N", "16"))
    w = synth.random_cae(seed=42)
    x = oracle.synth_crops(42, 0, n)
    ref = oracle.cae_forward(w, x, acc64=True, want=("features",), layers=True)
    s, t = w.bn_scale_shift()
    p1 = ref["layers"][0]
    f_ref = ref["features"].astype(np.float64)
    fmax = np.abs(f_ref).max()
    p2_ref = W.post_pool(W.conv_direct64(p1, w.kernels[1]), w.biases[1], s[1], t[1], np.float64)

    def feats(p2):
        return W.post_pool(W.conv_direct64(p2.astype(np.float32), w.kernels[2]), w.biases[2], s[2], t[2], np.float64).reshape(n, -1)
    f_base = feats(p2_ref)
    pts = (0, 1, -1, 2, -2)
    AT, G, BT = W.toom_cook(4, 3, pts)
    G_ = np.array(G, dtype=np.float64)
    U = np.einsum("ra,abio,cb->rcio", G_, w.kernels[1].astype(np.float64), G_).astype(np.float32).reshape(36, 32, 64)
    N, H, Wd, Ci = p1.shape
    xp = np.zeros((N, H + 2, Wd + 2, Ci), np.float32)
    xp[:, 1:-1, 1:-1] = p1
    d = np.empty((N, 8, 8, 6, 6, Ci), np.float32)
    for i in range(6):
        for j in range(6):
            d[:, :, :, i, j] = xp[:, i:i + H:4, j:j + Wd:4][:, :8, :8]
    V = W.f32mat_apply(BT, W.f32mat_apply(BT, d, 3), 4).reshape(-1, 36, Ci)
    for mode in ("f32", "x3", "x2"):
        M = contract(V, U, mode).reshape(-1, 6, 6, 64)
        Y = W.f32mat_apply(AT, W.f32mat_apply(AT, M, 1), 2).reshape(N, 8, 8, 4, 4, 64).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, Wd, 64)
        p2 = W.post_pool(Y, w.biases[1], s[1], t[1], np.float32)
        e2 = float(np.abs(p2 - p2_ref).max() / np.abs(p2_ref).max())
        ef = float(np.abs(feats(p2) - f_base).max() / fmax)
        print(json.dumps(dict(contraction={"f32": "f32 MFMA (k-ordered fma chain)", "x3": "split bf16, 6 products", "x2": "split bf16, 3 products"}[mode],
                              bf16_mfma_products_per_f32_product={"f32": None, "x3": 6, "x2": 3}[mode],
                              p2_err_over_max=e2, feature_err_over_max_from_conv2=ef, bar=1e-5)), flush=True)


if __name__ == "__main__":
    main()
