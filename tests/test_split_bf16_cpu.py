"""CPU tests (numpy) of the arithmetic the split-bf16 kernels rest on (csrc/conv45_bf16x3.hip, conv_generic_x3.hip,
conv67_x3_kernel, scaler_pca_x3_kernel, P1 of conv12_fused.hip).  No GPU: these pin the algebra --
what the hardware adds to it is measured by the -m gpu tests."""
import numpy as np

from cellscreen import synth


def bf16(x):
    """round-to-nearest-even float32 -> bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 and the host packers do)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    a1 = bf16(x)
    r1 = (x - a1).astype(np.float32)
    a2 = bf16(r1)
    r2 = (r1 - a2).astype(np.float32)
    return a1, a2, bf16(r2)


def test_three_bf16_terms_hold_a_float32_to_its_last_bit_or_2_to_the_minus_24():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200_000).astype(np.float32), rng.uniform(0, 1, 200_000).astype(np.float32),
                        np.float32([0.0, 1.0, -1.0, 1e-30, 3e38, 0.1, 1 / 3])])
    a1, a2, a3 = split3(x)
    # every residual is exact in fp32 (what the kernels rely on when they subtract in fp32)
    assert np.array_equal((x.astype(np.float64) - a1).astype(np.float32).astype(np.float64), x.astype(np.float64) - a1)
    err = np.abs(x.astype(np.float64) - (a1.astype(np.float64) + a2 + a3))
    assert (err <= np.abs(x) * 2.0 ** -24).all()
    # round-to-nearest leaves most values exact; what is left is below half an ulp of the third term
    assert (err == 0).mean() > 0.5


def test_six_products_reach_fp32_accuracy_three_do_not():
    """K = 288 dot products (conv4's shape) from the split operands, partial products accumulated exactly per MFMA-sized block of
    32 and rounded to fp32 between blocks: six products beat the fp32 fma chain, three stay at 2^-16."""
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 1, (4096, 288)).astype(np.float32)
    b = (rng.uniform(-1, 1, (288,)) * 0.06).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    chain = np.zeros(len(a), np.float32)
    for k in range(288):
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * b[k]).astype(np.float32)
    A, B = split3(a), split3(b)

    def blocks(pairs):
        acc = np.zeros(len(a), np.float32)
        for k0 in range(0, 288, 32):
            for i, j in pairs:      # one MFMA per (block, pair): exact products, one rounding into the fp32 accumulator
                acc = (acc.astype(np.float64) + A[i][:, k0:k0 + 32].astype(np.float64) @ B[j][k0:k0 + 32].astype(np.float64)).astype(np.float32)
        return acc

    scale = np.abs(ref).max()
    e_chain = np.abs(chain - ref).max() / scale
    e6 = np.abs(blocks([(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]) - ref).max() / scale
    e3 = np.abs(blocks([(0, 1), (1, 0), (0, 0)]) - ref).max() / scale
    assert e6 <= e_chain and e6 < 3e-7
    assert 3e-6 < e3 < 1e-4


def test_conv1_records_packed_along_k_are_the_six_products():
    """P1 of conv12_fused.hip: a pixel is the record [x1, x2, x3, x1]; against B = [w1,0,0,0], [w2,w1,0,0], [0,w2,w1,w3] per tap the
    three MFMAs give exactly x1w1 | x1w2 + x2w1 | x2w2 + x3w1 + x1w3; with the ninth tap in fp32 that is conv1 to fp32 accuracy
    (an ideal accumulator: what the MI355X adds when magnitudes are mixed in one instruction is measured on the GPU)."""
    w = synth.random_cae(seed=42)
    k = w.kernels[0][:, :, 0, :].astype(np.float32)                       # (3, 3, 32)
    x = synth.synth_crops(11, 7000, 6).astype(np.float32)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    taps = [(dy, dx) for dy in range(3) for dx in range(3)]
    win = lambda a, t: a[:, t[0]:t[0] + 64, t[1]:t[1] + 64, None].astype(np.float64)      # noqa: E731
    ref = sum(win(xp, t) * k[t].astype(np.float64) for t in taps)
    x1, x2, x3 = split3(xp)
    w1, w2, w3 = split3(k)
    rec = [x1, x2, x3, x1]                                                 # the record's four slots
    tabs = [[w1, 0, 0, 0], [w2, w1, 0, 0], [0, w2, w1, w3]]
    acc = np.zeros(ref.shape, np.float32)
    for tab in reversed(tabs):                                             # smallest magnitude first, as the kernel issues them
        part = sum(win(rec[s], t) * tab[s][t].astype(np.float64) for t in taps[:8] for s in range(4) if not np.isscalar(tab[s]))
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    acc = (acc.astype(np.float64) + win(xp, taps[8]) * k[taps[8]].astype(np.float64)).astype(np.float32)
    chain = np.zeros(ref.shape, np.float32)
    for t in taps:
        chain = (chain.astype(np.float64) + win(xp, t) * k[t].astype(np.float64)).astype(np.float32)
    scale = np.abs(ref).max()
    e_rec, e_chain = np.abs(acc - ref).max() / scale, np.abs(chain - ref).max() / scale
    assert e_rec <= e_chain and e_rec < 1.5e-7, (e_rec, e_chain)
