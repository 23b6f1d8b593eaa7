#!/usr/bin/env python3
"""Numerical study (CPU, numpy): a TWO-term fp16 split (3 matrix products) against the three-term bf16 split (6 products).

VERDICT r02 item 1.  fp16 carries 11 significant bits, so a = a_hi + a_lo (a_hi = fp16(a), a_lo = fp16(a - a_hi)) holds 22 of
the 24 bits of a float32, and a b ~ a_hi b_hi + (a_hi b_lo + a_lo b_hi): three products, each exact in the MFMA's fp32
accumulator, at the bf16 rate (v_mfma_f32_16x16x32_f16).  That is half the matrix instructions of the six-product bf16 form of
DESIGN.md section 3g and a 3-instruction split instead of 5.5 -- but fp16's exponent range is 2^-14 .. 2^15 (subnormals to
2^-24), so operands need exact power-of-two pre-scales, and the dropped a_lo b_lo term plus the 2^-22 operand representation
make it 4x coarser per operand than float32.  This script MEASURES what that does, in the form the kernels would run:

    v   = S a                      (S a power of two chosen from a BOUND on |a|: looseness L = bound / actual max)
    hi  = fp16(v)                  lo = fp16((v - hi) 2^11)          (the residual pre-scaled: same binades as hi, no subnormals)
    acc_hi += hi_a hi_b            acc_lo += hi_a lo_b + lo_a hi_b    (one MFMA per product and 32-deep K block)
    out = (acc_hi + 2^-11 acc_lo) / (S_a S_b)

on (1) conv2 as Winograd F(4x4,3x3) (the contraction M = V U only; transforms stay fp32), (2) conv6 in the folded-direct form
of conv67_x3_kernel, (3) the PCA GEMM, (4) conv3 as Winograd F(2x2,3x3) -- each against the float64 evaluation of the same
stage inputs and against the bars of tests/helpers.py, next to the fp32 fma chain and the six-product bf16 form.
'flush' rows model a matrix core that flushes fp16 subnormal INPUTS to zero (tests/test_gpu_fp16_mfma.py measures what the
MI355X does).  Decides nothing in the product by itself.  Lives under tests/ because it uses the CPU oracle.

    python tests/study_split_fp16.py            [N=16]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import study_wino_error as W  # noqa: E402
import helpers  # noqa: E402,F401
from cellscreen import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16(x, parts=3):
    out, rem = [], x.astype(np.float32)
    for _ in range(parts):
        p = bf16(rem)
        out.append(p)
        rem = (rem - p).astype(np.float32)
    return out


def fp16(x, flush=False):
    """float32 -> fp16 (round to nearest even, subnormals kept unless flush) -> float32"""
    with np.errstate(over="ignore"):
        h = np.asarray(x, np.float32).astype(np.float16)
    if flush:
        h = np.where(np.abs(h.astype(np.float32)) < 2.0 ** -14, np.float16(0), h)
    return h.astype(np.float32)


def pow2_scale(bound):
    """largest power of two S with S * bound <= 2^15 (fp16 tops out at 65504)"""
    return 2.0 ** (15 - int(np.ceil(np.log2(bound))))


def split_fp16(x, S, flush=False, prescale_lo=True):
    v = (x.astype(np.float32) * np.float32(S)).astype(np.float32)
    hi = fp16(v, flush)
    r = (v - hi).astype(np.float32)                         # exact in fp32
    lo = fp16(r * np.float32(2.0 ** 11 if prescale_lo else 1.0), flush)
    assert np.isfinite(hi).all() and np.isfinite(lo).all(), "fp16 overflow: scale too large"
    return hi, lo


def blockdot(A, B):
    """exact dot products of one 32-deep K block, as float64: [..., k] x [k, n]"""
    return A.astype(np.float64) @ B.astype(np.float64)


def gemm(A, B, mode, La=1.0, Lb=1.0):
    """C[m, n] = sum_k A[m, k] B[k, n] the way a kernel of the named arithmetic would: K walked in blocks of 32 (4 for the fp32
    MFMA's fma chain, modelled per term), each matrix instruction one exact block sum rounded once into its fp32 accumulator."""
    M, K = A.shape
    N = B.shape[1]
    if mode == "f64":
        return blockdot(A, B)
    if mode == "f32":
        acc = np.zeros((M, N), np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + A[:, k, None].astype(np.float64) * B[None, k, :].astype(np.float64)).astype(np.float32)
        return acc
    if mode == "bf16x3":
        As, Bs = split_bf16(A), split_bf16(B)
        hi = np.zeros((M, N), np.float32); lo = hi.copy(); lw = hi.copy()
        for k0 in range(0, K, 32):
            s = slice(k0, k0 + 32)
            hi = (hi + blockdot(As[0][:, s], Bs[0][s])).astype(np.float32)
            for i, j in ((0, 1), (1, 0)):
                lo = (lo + blockdot(As[i][:, s], Bs[j][s])).astype(np.float32)
            for i, j in ((1, 1), (0, 2), (2, 0)):
                lw = (lw + blockdot(As[i][:, s], Bs[j][s])).astype(np.float32)
        return (hi + (lo + lw).astype(np.float32)).astype(np.float32)
    return gemm_scaled(A, B, mode, np.abs(A).max() * La, np.abs(B).max() * Lb)


MODES = ("f32", "bf16x3", "fp16x2", "fp16x2 L=2^8", "fp16x2 L=2^16", "fp16x2 flush", "fp16x2 flush L=2^8", "fp16x2 onechain", "fp16x2x4")


def run_mode(A, B, mode):
    La = 1.0
    m = mode
    if "L=2^" in mode:
        La = 2.0 ** int(mode.split("L=2^")[1])
        m = mode.split(" L=")[0]
    return gemm(A, B, m, La, La)


def main():
    n = int(os.environ.get("N", "8"))
    w = synth.random_cae(seed=42)
    x = oracle.synth_crops(42, 0, n)
    x[n // 2:] = synth.blob_crops(3, n - n // 2)
    ref = oracle.cae_forward(w, x, acc64=True, want=("features", "recon", "mse"), layers=True)
    s, t = w.bn_scale_shift()
    L = ref["layers"]
    rows = []

    # ---- (1) conv2 as F(4x4,3x3): M[xi] = V[xi] U[xi]
    p1 = L[0]
    p2_ref = W.post_pool(W.conv_direct64(p1, w.kernels[1]), w.biases[1], s[1], t[1], np.float64)
    f_ref = W.post_pool(W.conv_direct64(p2_ref.astype(np.float32), w.kernels[2]), w.biases[2], s[2], t[2], np.float64).reshape(n, -1)
    fmax = np.abs(f_ref).max()
    AT, G, BT = W.toom_cook(4, 3, (0, 1, -1, 2, -2))
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
    print(json.dumps(dict(note="conv2 F(4x4,3x3) operand ranges", p1_max=float(np.abs(p1).max()), V_max=float(np.abs(V).max()),
                          U_max=float(np.abs(U).max()), U_min_nonzero=float(np.abs(U[U != 0]).min()))), flush=True)
    for mode in MODES:
        # one global scale for V and one for U (per-point scales would have to be undone before the output transform)
        Vall = V.transpose(1, 0, 2)                      # [36][tiles][ci]
        if mode in ("f32", "bf16x3"):
            M = np.stack([run_mode(Vall[q], U[q], mode) for q in range(36)], 1)
        else:
            # the scale is global: emulate by running the 36 points as one call with shared scales
            La = 2.0 ** int(mode.split("L=2^")[1]) if "L=2^" in mode else 1.0
            m = mode.split(" L=")[0]
            Sa = np.abs(V).max() * La
            Sb = np.abs(U).max() * La
            M = np.stack([gemm_scaled(Vall[q], U[q], m, Sa, Sb) for q in range(36)], 1)
        M = M.reshape(-1, 6, 6, 64)
        Y = W.f32mat_apply(AT, W.f32mat_apply(AT, M, 1), 2).reshape(N, 8, 8, 4, 4, 64).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, Wd, 64)
        p2 = W.post_pool(Y, w.biases[1], s[1], t[1], np.float32)
        e2 = float(np.abs(p2 - p2_ref).max() / np.abs(p2_ref).max())
        f = W.post_pool(W.conv_direct64(p2.astype(np.float32), w.kernels[2]), w.biases[2], s[2], t[2], np.float64).reshape(n, -1)
        ef = float(np.abs(f - f_ref).max() / fmax)
        rows.append(dict(stage="conv2 F(4x4,3x3)", contraction=mode, p2_err_over_max=e2, feature_err_over_max=ef, bars=dict(p2=3e-6, features=1e-5)))
        print(json.dumps(rows[-1]), flush=True)

    # ---- (4) conv3 as F(2x2,3x3) on the oracle's p2
    p2o = L[1]
    z3_ref = W.conv_direct64(p2o, w.kernels[2])
    f3_ref = W.post_pool(z3_ref, w.biases[2], s[2], t[2], np.float64).reshape(n, -1)
    AT2, G2, BT2 = W.toom_cook(2, 3, (0, 1, -1))
    G2_ = np.array(G2, dtype=np.float64)
    U3 = np.einsum("ra,abio,cb->rcio", G2_, w.kernels[2].astype(np.float64), G2_).astype(np.float32).reshape(16, 64, 32)
    N, H, Wd, Ci = p2o.shape
    xp = np.zeros((N, H + 2, Wd + 2, Ci), np.float32)
    xp[:, 1:-1, 1:-1] = p2o
    d = np.empty((N, H // 2, Wd // 2, 4, 4, Ci), np.float32)
    for i in range(4):
        for j in range(4):
            d[:, :, :, i, j] = xp[:, i:i + H:2, j:j + Wd:2][:, :H // 2, :Wd // 2]
    V3 = W.f32mat_apply(BT2, W.f32mat_apply(BT2, d, 3), 4).reshape(-1, 16, Ci).transpose(1, 0, 2)
    for mode in MODES:
        if mode in ("f32", "bf16x3"):
            M = np.stack([run_mode(V3[q], U3[q], mode) for q in range(16)], 1)
        else:
            La = 2.0 ** int(mode.split("L=2^")[1]) if "L=2^" in mode else 1.0
            m = mode.split(" L=")[0]
            M = np.stack([gemm_scaled(V3[q], U3[q], m, np.abs(V3).max() * La, np.abs(U3).max() * La) for q in range(16)], 1)
        M = M.reshape(-1, 4, 4, 32)
        Y = W.f32mat_apply(AT2, W.f32mat_apply(AT2, M, 1), 2).reshape(N, H // 2, Wd // 2, 2, 2, 32).transpose(0, 1, 3, 2, 4, 5).reshape(N, H, Wd, 32)
        f3 = W.post_pool(Y, w.biases[2], s[2], t[2], np.float32).reshape(n, -1)
        rows.append(dict(stage="conv3 F(2x2,3x3)", contraction=mode, feature_err_over_max=float(np.abs(f3 - f3_ref).max() / np.abs(f3_ref).max()),
                         bars=dict(features=1e-5)))
        print(json.dumps(rows[-1]), flush=True)

    # ---- (2) conv6, folded direct: four phase convs of 2x2 taps over the stored a5 grid (16x16x64 -> 32x32x32)
    a5 = L[4]
    k6 = w.kernels[5].astype(np.float64)
    up = np.repeat(np.repeat(a5, 2, axis=1), 2, axis=2)
    z6_ref = W.conv_direct64(up, w.kernels[5])
    a6_ref = (np.maximum(z6_ref + w.biases[5], 0) * s[5] + t[5])
    N, Hs, Ws, Ci = a5.shape
    ap = np.zeros((N, Hs + 2, Ws + 2, Ci), np.float32)
    ap[:, 1:-1, 1:-1] = a5
    # W_eff[a][b][ry][rx] = sum of the taps (dy, dx) that land on stored offset (ry, rx) for phase (a, b): rows {0: [(0,), (1,2)], 1: [(0,1), (2,)]}
    grp = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    for mode in MODES:
        z = np.zeros((N, 2 * Hs, 2 * Ws, 32), np.float32)
        for a in range(2):
            for b in range(2):
                Aop = np.concatenate([ap[:, a + ry:a + ry + Hs, b + rx:b + rx + Ws] for ry in range(2) for rx in range(2)], -1).reshape(-1, 4 * Ci)
                Bop = np.concatenate([sum(k6[dy, dx] for dy in grp[a][ry] for dx in grp[b][rx]) for ry in range(2) for rx in range(2)], 0).astype(np.float32)
                if mode in ("f32", "bf16x3"):
                    zz = run_mode(Aop, Bop, mode)
                else:
                    La = 2.0 ** int(mode.split("L=2^")[1]) if "L=2^" in mode else 1.0
                    zz = gemm_scaled(Aop, Bop, mode.split(" L=")[0], np.abs(a5).max() * La, np.abs(k6).sum(axis=(0, 1)).max() * La)
                z[:, a::2, b::2] = zz.reshape(N, Hs, Ws, 32)
        a6 = (np.maximum(z + w.biases[5], 0) * s[5] + t[5]).astype(np.float32)
        # what conv7 + sigmoid + MSE make of it (float64 from here on): the bar is 1e-5 relative on the per-cell MSE
        def mse_of(a6x):
            u7 = np.repeat(np.repeat(a6x.astype(np.float64), 2, axis=1), 2, axis=2)
            r = 1.0 / (1.0 + np.exp(-(W.conv_direct64(u7, w.kernels[6])[..., 0] + float(w.biases[6][0]))))
            return ((x.astype(np.float64) - r) ** 2).mean(axis=(1, 2)), r
        m_ref, r_ref = mse_of(a6_ref)
        m_got, r_got = mse_of(a6)
        rows.append(dict(stage="conv6 folded direct", contraction=mode, a6_err_over_max=float(np.abs(a6 - a6_ref).max() / np.abs(a6_ref).max()),
                         recon_abs_err=float(np.abs(r_got - r_ref).max()), mse_rel_err=float((np.abs(m_got - m_ref) / m_ref).max()),
                         bars=dict(recon=1e-5, mse=1e-5)))
        print(json.dumps(rows[-1]), flush=True)

    # ---- (3) the PCA GEMM on scaled features
    g = np.load(os.path.join(HERE, "golden", "golden_detector.npz"))
    det = helpers.det_from_golden(g)
    feats = ref["features"]
    scaled = ((feats.astype(np.float64) - det.scaler_center.astype(np.float64)) / det.scaler_scale.astype(np.float64)).astype(np.float32)
    comps = det.pca_components.astype(np.float32)
    pca_ref = scaled.astype(np.float64) @ comps.astype(np.float64).T
    for mode in MODES:
        got = run_mode(scaled, np.ascontiguousarray(comps.T), mode)
        rows.append(dict(stage="pca gemm", contraction=mode, pca_err_over_max=float(np.abs(got - pca_ref).max() / np.abs(pca_ref).max()),
                         scaled_max=float(np.abs(scaled).max()), bars=dict(pca=1e-5)))
        print(json.dumps(rows[-1]), flush=True)


def gemm_scaled(A, B, mode, bound_a, bound_b):
    """gemm() with the operand bounds given (a scale shared by several calls)"""
    M, K = A.shape
    N = B.shape[1]
    flush = "flush" in mode
    pre = "onechain" not in mode
    Sa, Sb = pow2_scale(bound_a), pow2_scale(bound_b)
    ah, al = split_fp16(A, Sa, flush, pre)
    bh, bl = split_fp16(B, Sb, flush, pre)
    hi = np.zeros((M, N), np.float32); lo = hi.copy(); lw = hi.copy()
    for k0 in range(0, K, 32):
        s = slice(k0, k0 + 32)
        hi = (hi + blockdot(ah[:, s], bh[s])).astype(np.float32)
        tgt = lo if pre else hi
        tgt = (tgt + blockdot(ah[:, s], bl[s])).astype(np.float32)
        tgt = (tgt + blockdot(al[:, s], bh[s])).astype(np.float32)
        if pre:
            lo = tgt
        else:
            hi = tgt
        if "x4" in mode:
            lw = (lw + blockdot(al[:, s], bl[s])).astype(np.float32)
    if pre:
        if "x4" in mode:
            lo = (lo + lw * np.float32(2.0 ** -11)).astype(np.float32)
        out = (hi + lo * np.float32(2.0 ** -11)).astype(np.float32)
    else:
        out = hi
    return (out * np.float32(1.0 / (Sa * Sb))).astype(np.float32)


if __name__ == "__main__":
    main()
