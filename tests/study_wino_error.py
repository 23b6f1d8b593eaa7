#!/usr/bin/env python3
"""Numerical study (CPU, numpy): what would Winograd F(m x m', 3x3) with larger tiles cost conv2 in accuracy?

VERDICT r01 item 3: evaluate F(4x4,3x3) for conv2 (CAE_improved_modeltrain.py:195-197) by MEASURING the
feature / score error before adopting.  The kernels' arithmetic is emulated in float32: transforms as
sequences of fp32 adds / multiplies in matrix order, the channel contraction as the MFMA's k-ordered fp32
fma chain, U = G g G^T evaluated in double and rounded once (what the host packers do).

Everything is compared with the fp64-evaluated oracle at the bars of tests/helpers.py:
    features  max|err| <= 1e-5 * max|f|          scores  |err| <= 1e-4 * sum|alpha|
Prints one JSON line per variant.  Lives under tests/ because it uses the CPU oracle as its checker (only tests/,
smoke() and bench.py's cpu_baseline leg may); pytest does not collect it (not a test_*.py file):
    python tests/study_wino_error.py          [N=48 BLOBS=1 ONLY=F(4x4)]
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from cellscreen import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def toom_cook(m, r, pts):
    """A^T (m x n), G (n x r), B^T (n x n) of F(m, r) for the finite points `pts` (+ infinity), as Fractions:
    y = A^T [(G g) . (B^T d)].  B^T = C^-T with C the evaluation matrix of degree n-1 polynomials; rows are
    rescaled so that B^T is integral where the standard forms are."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fraction(p) for p in pts]

    def ev(deg):
        rows = [[p ** k for k in range(deg)] for p in pts]
        rows.append([Fraction(0)] * (deg - 1) + [Fraction(1)])
        return rows
    AT = [list(col) for col in zip(*ev(m))]
    G = ev(r)
    Cm = ev(n)
    # invert C (Gauss-Jordan over Fractions)
    a = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(Cm)]
    for c in range(n):
        piv = next(i for i in range(c, n) if a[i][c] != 0)
        a[c], a[piv] = a[piv], a[c]
        pv = a[c][c]
        a[c] = [v / pv for v in a[c]]
        for i in range(n):
            if i != c and a[i][c] != 0:
                f = a[i][c]
                a[i] = [vi - f * vc for vi, vc in zip(a[i], a[c])]
    Cinv = [row[n:] for row in a]
    BT = [list(col) for col in zip(*Cinv)]
    # scale row j of B^T by N_j = prod_{k != j}(a_j - a_k) and row j of G by 1/N_j
    for j in range(n - 1):
        N = Fraction(1)
        for k in range(n - 1):
            if k != j:
                N *= pts[j] - pts[k]
        BT[j] = [v * N for v in BT[j]]
        G[j] = [v / N for v in G[j]]
    return AT, G, BT


def f32mat_apply(M, x, axis):
    """y = M x along `axis` as a VALU kernel would: fp32 products accumulated left to right, zero entries
    skipped, +-1 entries as adds."""
    x = np.moveaxis(x, axis, 0)
    out = []
    for row in M:
        acc = None
        for c, xv in zip(row, x):
            c = float(c)
            if c == 0.0:
                continue
            term = xv if c == 1.0 else (-xv if c == -1.0 else (np.float32(c) * xv).astype(np.float32))
            acc = term if acc is None else (acc + term).astype(np.float32)
        out.append(acc if acc is not None else np.zeros_like(x[0]))
    return np.moveaxis(np.stack(out, 0), 0, axis)


def conv_wino(p_in, k_hwio, mh, mw, pts_h, pts_w):
    """'same' 3x3 conv of NHWC fp32 `p_in` by F(mh x mw, 3x3); returns fp32 NHWC (no bias)."""
    n, H, W, Ci = p_in.shape
    Co = k_hwio.shape[3]
    ATh, Gh, BTh = toom_cook(mh, 3, pts_h)
    ATw, Gw, BTw = toom_cook(mw, 3, pts_w)
    nh, nw = mh + 2, mw + 2
    # U = G g G^T in double, rounded once
    Gh_ = np.array(Gh, dtype=np.float64)
    Gw_ = np.array(Gw, dtype=np.float64)
    U = np.einsum("ra,abio,cb->rcio", Gh_, k_hwio.astype(np.float64), Gw_).astype(np.float32)
    xp = np.zeros((n, H + 2, W + 2, Ci), np.float32)
    xp[:, 1:-1, 1:-1] = p_in
    th, tw = H // mh, W // mw
    # patches [n, th, tw, nh, nw, Ci]
    d = np.empty((n, th, tw, nh, nw, Ci), np.float32)
    for i in range(nh):
        for j in range(nw):
            d[:, :, :, i, j] = xp[:, i:i + H:mh, j:j + W:mw][:, :th, :tw]
    V = f32mat_apply(BTh, d, 3)
    V = f32mat_apply(BTw, V, 4)
    # M[xi] = V[xi] U[xi] : k-ordered fp32 fma chain (fma = exact product + one rounding, via float64)
    Vf = V.reshape(-1, nh, nw, Ci)
    M = np.zeros(Vf.shape[:3] + (Co,), np.float32)
    for k in range(Ci):
        M = (M.astype(np.float64) + Vf[..., k, None].astype(np.float64) * U[None, :, :, k, :].astype(np.float64)).astype(np.float32)
    Y = f32mat_apply(ATh, M, 1)
    Y = f32mat_apply(ATw, Y, 2)          # [tiles, mh, mw, Co]
    Y = Y.reshape(n, th, tw, mh, mw, Co).transpose(0, 1, 3, 2, 4, 5).reshape(n, H, W, Co)
    return Y


def conv_direct64(p_in, k_hwio):
    n, H, W, Ci = p_in.shape
    xp = np.zeros((n, H + 2, W + 2, Ci), np.float64)
    xp[:, 1:-1, 1:-1] = p_in
    out = np.zeros((n, H, W, k_hwio.shape[3]), np.float64)
    k = k_hwio.astype(np.float64)
    for a in range(3):
        for b in range(3):
            out += np.einsum("nhwi,io->nhwo", xp[:, a:a + H, b:b + W], k[a, b])
    return out


def post_pool(z, bias, s, t, dtype):
    v = (z + bias.astype(dtype)).astype(dtype)
    v = np.maximum(v, 0)
    v = (v * s.astype(dtype) + t.astype(dtype)).astype(dtype)
    n, H, W, C = v.shape
    return v.reshape(n, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def main():
    n = int(os.environ.get("N", "48"))
    w = synth.random_cae(seed=42)
    x = oracle.synth_crops(42, 0, n)
    if os.environ.get("BLOBS"):
        x = synth.blob_crops(3, n)
    ref = oracle.cae_forward(w, x, acc64=True, want=("features",), layers=True)
    s, t = w.bn_scale_shift()
    p1 = ref["layers"][0]                       # fp32 rounding of the fp64-evaluated p1
    f_ref = ref["features"].astype(np.float64)
    p2_ref64 = post_pool(conv_direct64(p1, w.kernels[1]), w.biases[1], s[1], t[1], np.float64)
    print(json.dumps(dict(check_p2_vs_oracle=float(np.abs(p2_ref64 - ref["layers"][1]).max() / np.abs(p2_ref64).max()))))

    def features_from_p2(p2):   # conv3 in double on the given p2: isolates what conv2's error does to the features
        z = conv_direct64(p2.astype(np.float32), w.kernels[2])
        return post_pool(z, w.biases[2], s[2], t[2], np.float64).reshape(n, -1)
    f_base = features_from_p2(p2_ref64)
    fmax = np.abs(f_ref).max()
    variants = {
        "F(2x2) pts(0,1,-1)": (2, 2, (0, 1, -1), (0, 1, -1)),
        "F(4x4) pts(0,1,-1,2,-2)": (4, 4, (0, 1, -1, 2, -2), (0, 1, -1, 2, -2)),
        "F(4x4) pts(0,1,-1,1/2,-1/2)": (4, 4, (0, 1, -1, Fraction(1, 2), Fraction(-1, 2)),) * 0 or (4, 4, (0, 1, -1, Fraction(1, 2), Fraction(-1, 2)), (0, 1, -1, Fraction(1, 2), Fraction(-1, 2))),
        "F(4x4) pts(0,1,-1,1/2,-2)": (4, 4, (0, 1, -1, Fraction(1, 2), -2), (0, 1, -1, Fraction(1, 2), -2)),
        "F(4x4) pts(0,1,-1,2,-1/2)": (4, 4, (0, 1, -1, 2, Fraction(-1, 2)), (0, 1, -1, 2, Fraction(-1, 2))),
        "F(2x4) pts w(0,1,-1,2,-2)": (2, 4, (0, 1, -1), (0, 1, -1, 2, -2)),
        "F(2x4) pts w(0,1,-1,1/2,-1/2)": (2, 4, (0, 1, -1), (0, 1, -1, Fraction(1, 2), Fraction(-1, 2))),
        "F(2x4) pts w(0,1,-1,1/2,-2)": (2, 4, (0, 1, -1), (0, 1, -1, Fraction(1, 2), -2)),
    }
    only = os.environ.get("ONLY")
    for name, (mh, mw, ph, pw) in variants.items():
        if only and only not in name:
            continue
        z = conv_wino(p1, w.kernels[1], mh, mw, ph, pw)
        p2 = post_pool(z, w.biases[1], s[1], t[1], np.float32)
        e2 = float(np.abs(p2 - p2_ref64).max() / np.abs(p2_ref64).max())
        f = features_from_p2(p2)
        ef = float(np.abs(f - f_base).max() / fmax)
        print(json.dumps(dict(variant=name, mults_per_output=round((mh + 2) * (mw + 2) / (mh * mw), 3),
                              p2_err_over_max=e2, feature_err_over_max_from_conv2=ef, bar=1e-5)), flush=True)


if __name__ == "__main__":
    main()
