"""BASELINE.json configs[4] as a parity case: the reference's layer grammar at another size --
128x128 crops, filters 32-64-128 | 128-64-32-1 (create_improved_autoencoder
takes input_shape, CAE_improved_modeltrain.py:184) -- through the same C ABI, on the run-time-shaped
MFMA kernels of csrc/conv_generic.hip, against the CPU oracle (fp64 accumulation) at the tolerances
of the reference architecture (helpers.py).  Plus a rectangular variant and the refusal paths."""
import numpy as np
import pytest

import helpers as H
from cellscreen import _lib as L
from cellscreen import spec, synth
from cellscreen.detector_fit import fit_detector
from cellscreen.engine import Engine
from oracle import oracle

pytestmark = pytest.mark.gpu

LARGE_HW = (128, 128)
LARGE_CH = (32, 64, 128, 128, 64, 32, 1)


@pytest.fixture(scope="module")
def large():
    return synth.random_cae(seed=5, hw=LARGE_HW, channels=LARGE_CH, n_enc=3)


@pytest.fixture(scope="module")
def crops():
    return np.concatenate([synth.synth_crops(42, 0, 6, hw=LARGE_HW), synth.blob_crops(4, 6, hw=LARGE_HW)])


def test_large_variant_layers_and_reconstruction(large, crops):
    e = Engine.from_weights(large)
    try:
        # automatic pass size for device-resident input: what a ~28 GB workspace holds (1.9 MB of activations per cell here)
        assert e.info.chunk_cells == 14336
        assert (e.info.height, e.info.width, e.info.n_conv, e.info.n_enc) == (128, 128, 7, 3)
        assert e.info.reference_arch == 0 and e.info.feature_dim == 16 * 16 * 128
        assert list(e.info.channels[:7]) == list(LARGE_CH)
        ref = oracle.cae_forward(large, crops, acc64=True, layers=True)
        for l in range(7):
            got = e.layer_output(crops, l)
            want = ref["layers"][l].reshape(got.shape)
            H.assert_close_scaled(got, want, H.TOL_FEATURES if l < 6 else H.TOL_RECON, f"layer {l}")
        rec, mse, mae = e.reconstruct(crops)
        assert np.abs(rec - ref["recon"].reshape(rec.shape)).max() <= H.TOL_RECON
        H.assert_rel(mse, ref["mse"], H.TOL_ERR_REL, "mse")
        H.assert_rel(mae, ref["mae"], H.TOL_ERR_REL, "mae")
        f = e.encode(crops)
        assert f.shape == (len(crops), 32768)
        H.assert_close_scaled(f, ref["features"].reshape(f.shape), H.TOL_FEATURES, "features (h,w,c)")
    finally:
        e.close()


def test_large_variant_screen_end_to_end(large):
    """Detector fitted (real scikit-learn) on GPU-encoded features of the large model; scores and flags
    against the oracle pipeline."""
    train = synth.blob_crops(11, 160, hw=LARGE_HW)
    e0 = Engine.from_weights(large)
    feats = e0.encode(train)
    e0.close()
    det, _ = fit_detector(feats, pca_random_state=0)
    assert det.n_features == 32768
    x = np.concatenate([synth.blob_crops(12, 20, hw=LARGE_HW), synth.synth_crops(7, 0, 4, hw=LARGE_HW)])
    e = Engine.from_weights(large, None, det)
    try:
        r = e.screen(x)
        ref = oracle.screen(large, None, det, x, acc64=True)
        H.assert_rel(r["mse"], ref["mse"], H.TOL_ERR_REL, "mse")
        for name, p in (("cons", det.conservative), ("mod", det.moderate)):
            tol = H.TOL_DEC_E2E * np.abs(p.dual_coef).sum()
            assert np.abs(r[f"{name}_score"] - ref[f"{name}_score"]).max() <= tol, name
            H.flags_agree(-r[f"{name}_score"], r[f"{name}_pred"], ref[f"{name}_dec"], ref[f"{name}_pred"], tol, name)
        # chunking and device-resident input give the same bits
        import torch
        e.set_chunk(7)
        r2 = e.screen(torch.from_numpy(x).cuda())
        assert np.array_equal(r2["mse"].cpu().numpy(), r["mse"]) and np.array_equal(r2["mod_score"].cpu().numpy(), r["mod_score"])
    finally:
        e.close()


def test_rectangular_variant_with_odd_channel_counts():
    w = synth.random_cae(seed=9, hw=(64, 128), channels=(16, 32, 48, 48, 32, 16, 1), n_enc=3)
    x = synth.synth_crops(3, 0, 5, hw=(64, 128))
    e = Engine.from_weights(w)
    try:
        ref = oracle.cae_forward(w, x, acc64=True)
        rec, mse, mae = e.reconstruct(x)
        assert np.abs(rec - ref["recon"].reshape(rec.shape)).max() <= H.TOL_RECON
        H.assert_rel(mse, ref["mse"], H.TOL_ERR_REL, "mse")
        H.assert_close_scaled(e.encode(x), ref["features"].reshape(5, -1), H.TOL_FEATURES, "features")
    finally:
        e.close()


@pytest.mark.parametrize("scale", [1.0, 255.0, 1.0e-3])
def test_generic_fp16_split_convs_scale_with_the_data(scale):
    """precision="split16": the run-time-shaped convs take their contraction as a two-term fp16 split (conv_generic_x3_kernel<.., H2>: a
    strip is staged as fp32, its own max|x| fixes a power-of-two scale, then it is split in place).  Every layer against the fp64 oracle
    at the unchanged tolerance, for crops in [0, 1], raw 8-bit values and values of 1e-3 (fp16's range is 2^-24 .. 65,504: an unscaled
    split would overflow or vanish); three 16-bit matrix instructions per (16 px, 16 filters, 32 channels); and a cell's result does
    not depend on what it is screened with."""
    hw = (64, 128)
    w = synth.random_cae(seed=13, hw=hw, channels=(32, 64, 128, 128, 64, 32, 1), n_enc=3)
    x = (np.concatenate([synth.synth_crops(5, 0, 3, hw=hw), synth.blob_crops(6, 3, hw=hw)]) * np.float32(scale)).astype(np.float32)
    ref = oracle.cae_forward(w, x, acc64=True, layers=True)["layers"]
    names = ["conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv3_relu_bn_pool", "conv4_relu_bn", "conv5_up_relu_bn", "conv6_up_relu_bn"]
    e = Engine.from_weights(w)
    prof = e.profile()
    got = [e.layer_output(x, l) for l in range(6)]
    alone = e.layer_output(x[2:3], 5)
    e.close()
    rows = spec.layer_table(hw, (32, 64, 128, 128, 64, 32, 1), 3)
    for l in range(1, 6):
        gh, gw = rows[l]["conv_hw"]
        taps = 4.0 if l > 3 else 9.0                       # folded upsample: 16 (phase, tap) pairs over a quarter of the grid
        assert prof[names[l]]["bf16_mfma_per_cell"] == gh * gw / 16 * (rows[l]["cout"] // 16) * taps * (rows[l]["cin"] / 32) * 3, l
        assert prof[names[l]]["mfma_per_cell"] == 0
    for l in range(6):
        H.assert_close_scaled(got[l], ref[l].reshape(got[l].shape), H.TOL_FEATURES, f"layer {l}, fp16 split, input scale {scale:g}")
    assert np.array_equal(alone[0], got[5][2])


@pytest.mark.parametrize("channels,expect", [((32, 64, 128, 128, 64, 32, 1), {1, 2, 3, 4, 5}), ((32, 40, 32, 32, 40, 32, 1), {1, 4})])
def test_split16_convs_against_the_fp32_exact_kernels_and_the_oracle(channels, expect):
    """csrc/conv_generic_x3.hip: with precision="split16" the MFMA convs of a non-reference architecture take the fp32 contraction on
    the 16-bit matrix pipe where the layer's shape has a plan (cin 32 / 64 / 128; plain and folded-upsample forms; a filter count
    that is not a multiple of 16).  Every layer against the fp64 oracle at the unchanged tolerance, and against
    precision="fp32_exact" (the fp32-MFMA kernels): different bits where a split kernel ran, the same error class."""
    hw = (64, 128)
    w = synth.random_cae(seed=13, hw=hw, channels=channels, n_enc=3)
    x = np.concatenate([synth.synth_crops(5, 0, 3, hw=hw), synth.blob_crops(6, 3, hw=hw)])
    ref = oracle.cae_forward(w, x, acc64=True, layers=True)["layers"]
    names = ["conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv3_relu_bn_pool", "conv4_relu_bn", "conv5_up_relu_bn", "conv6_up_relu_bn"]
    e = Engine.from_weights(w)
    prof = e.profile()
    on = {l for l in range(6) if prof[names[l]]["bf16_mfma_per_cell"] > 0}
    assert on >= expect, (on, expect)
    assert all(prof[names[l]]["mfma_per_cell"] == 0 for l in on) and all(prof[names[l]]["mfma_per_cell"] > 0 for l in set(range(6)) - on)
    got = [e.layer_output(x, l) for l in range(7)]
    e.close()
    e = Engine.from_weights(w, precision="fp32_exact")
    assert e.precision == "fp32_exact" and all(v["bf16_mfma_per_cell"] == 0 for v in e.profile().values())
    base = [e.layer_output(x, l) for l in range(7)]
    e.close()
    errs = {}
    for l in range(7):
        want = ref[l].reshape(got[l].shape)
        ea = H.assert_close_scaled(got[l], want, H.TOL_FEATURES if l < 6 else H.TOL_RECON, f"layer {l}, split16")
        eb = H.assert_close_scaled(base[l], want, H.TOL_FEATURES if l < 6 else H.TOL_RECON, f"layer {l}, fp32_exact")
        errs[l] = (float(f"{ea:.2e}"), float(f"{eb:.2e}"))
        if l in on:
            assert not np.array_equal(got[l], base[l]), l
    assert np.array_equal(got[0], base[0])              # conv1 (cin = 1) is the same kernel either way
    print("layer: (split16, fp32_exact) max err / max|ref| vs the fp64 oracle:", errs, "split16 layers:", sorted(on))


def test_separate_encoder_on_the_generic_path(large, crops):
    enc = synth.perturbed_encoder(large)
    e = Engine.from_weights(large, enc)
    try:
        assert e.info.shared_encoder == 0
        f1 = e.encode(crops[:4], which=1)
        ref = oracle.cae_forward(enc, crops[:4], acc64=True, want=("features",))
        H.assert_close_scaled(f1, ref["features"].reshape(4, -1), H.TOL_FEATURES, "encoder.keras features")
        f0 = e.encode(crops[:4], which=0)
        assert not np.array_equal(f0, f1)
    finally:
        e.close()


def test_unsupported_shapes_are_refused():
    with pytest.raises(L.CellScreenError) as ei:                 # conv4 grid 8 wide: below the generic kernel's 16
        Engine.from_weights(synth.random_cae(seed=1, hw=(64, 64), channels=(16, 32, 64, 64, 32, 16, 1), n_enc=3))
    assert ei.value.status == -6
    with pytest.raises(L.CellScreenError) as ei:                 # not the reference grammar
        Engine.from_weights(synth.random_cae(seed=1, hw=(64, 64), channels=(32, 64, 32, 32, 1), n_enc=3))
    assert ei.value.status == -6
    from cellscreen.trainer import Trainer
    with pytest.raises(L.CellScreenError) as ei:                 # the BatchNormalization kernels need filter counts dividing 256
        Trainer(synth.random_cae(seed=5, hw=(64, 128), channels=(48, 48, 48, 48, 48, 48, 1), n_enc=3))
    assert ei.value.status == -6


# ---- training on the run-time-shaped kernels (csrc/train_generic.hip) ------------------------------------------------
def _activation_pattern(tr, w, n):
    """The trainer's ReLU masks and max-pool routing from its relu outputs (stage tap 0), for any instance of the grammar
    (see tests/test_gpu_train.py: fp32 and fp64 disagree on a handful of those discontinuous decisions)."""
    nl, ne = w.n_conv - 1, w.n_enc
    masks, args = [], []
    for l in range(nl):
        r = tr.tensor(0, l, n)
        masks.append(r > 0)
        if l < ne:
            N, Hh, Ww, C = r.shape
            win = r.reshape(N, Hh // 2, 2, Ww // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(N, Hh // 2, Ww // 2, C, 4)
            args.append(np.argmax(win * np.sign(w.bn_gamma[l])[None, None, None, :, None], axis=-1))
        else:
            args.append(None)
    return masks + [None], args + [None]


@pytest.mark.parametrize("hw,channels,n", [((64, 128), (8, 16, 32, 32, 16, 8, 1), 5), (LARGE_HW, LARGE_CH, 2)])
def test_generic_trainer_gradients_against_the_oracle(hw, channels, n):
    """BASELINE.json configs[4]'s training half: forward (BN batch statistics) + backward of a non-reference instance of the
    layer grammar -- a small rectangular one and the 128x128 / 128-channel variant -- against oracle/train_oracle.py
    (numpy float64, pinned to torch autograd) on the trainer's own activation pattern, at the reference graph's bar."""
    from cellscreen.trainer import Trainer, param_layout, split_flat
    from oracle import train_oracle as T
    w = synth.random_cae(seed=13, hw=hw, channels=channels, n_enc=3)
    y = synth.blob_crops(17, n, hw=hw)
    x = np.clip(y + np.random.default_rng(3).normal(0, 0.02, y.shape), 0, 1).astype(np.float32)
    tr = Trainer(w)
    try:
        assert tr.n_trainable == w.n_params() - 2 * sum(c for c in channels[:-1])
        loss, mae = tr.forward_backward(x, y)
        masks, args = _activation_pattern(tr, w, n)
        st = T.TrainState(w, dtype=np.float64)
        ref = T.forward_backward(st, x, y, relu_masks=masks, pool_args=args)
        free = T.forward_backward(T.TrainState(w, dtype=np.float64), x, y)
        nl = w.n_conv - 1
        flips = sum(int(np.sum(m != (r > 0))) for m, r in zip(masks[:nl], free["relu"][:nl]))
        assert flips <= 1e-5 * sum(m.size for m in masks[:nl])
        assert abs(loss - ref["loss"]) <= 1e-5 * ref["loss"] and abs(mae - ref["mae"]) <= 1e-5 * ref["mae"]
        _, mov, g = tr.export_flat(grads=True)
        got = split_flat(g, param_layout(channels))
        errs = {}
        for (name, _shape), gr in zip(param_layout(channels), ref["grads"]):
            errs[name] = np.linalg.norm(got[name].astype(np.float64) - gr) / max(np.linalg.norm(gr), 1e-30)
        print("gradient relative L2 errors:", {k: float("%.2e" % v) for k, v in errs.items()})
        assert max(errs.values()) <= 1e-5, max(errs, key=errs.get)
        o = 0
        for l in range(nl):
            c = channels[l]
            assert np.allclose(mov[o:o + c], st.mov_mean[l], rtol=1e-5, atol=1e-7); o += c
            assert np.allclose(mov[o:o + c], st.mov_var[l], rtol=1e-5, atol=1e-7); o += c
    finally:
        tr.close()


def test_large_variant_trains_and_splits_for_the_gradient_all_reduce():
    """Trainer(large): the loss falls, evaluate() runs the inference graph, the exported weights load into the engine, and
    forward_backward -> (the all-reduce of the 1.34 MB flat gradient would go here) -> apply equals step bit for bit."""
    import torch
    from cellscreen.trainer import Trainer
    w = synth.random_cae(seed=5, hw=LARGE_HW, channels=LARGE_CH, n_enc=3, trivial_bn=True)
    X = torch.from_numpy(synth.blob_crops(31, 64, hw=LARGE_HW)).cuda()
    a, b = Trainer(w), Trainer(w)
    assert a.n_trainable == 334_593 - 2 * (32 + 64 + 128 + 128 + 64 + 32)     # SURVEY.md Appendix A.2 counts the moving statistics too
    g = torch.zeros(a.n_trainable, dtype=torch.float32, device="cuda")
    a.use_grad_tensor(g)
    losses = []
    for s in range(12):
        xb = X[torch.randint(0, 64, (8,), device="cuda")]
        la, _ = a.forward_backward(xb, xb)
        a.apply(1e-3)
        lb, _ = b.step(xb, xb, 1e-3)
        assert la == lb
        losses.append(la)
    assert losses[-1] < losses[0]
    pa, ma = a.export_flat()
    pb, mb = b.export_flat()
    assert np.array_equal(pa, pb) and np.array_equal(ma, mb) and float(g.abs().max()) > 0
    ev = a.evaluate(X[:16], X[:16])
    e = Engine.from_weights(a.weights())
    _, mse, _ = e.reconstruct(X[:16].cpu().numpy(), want_recon=False)
    assert abs(float(mse.mean()) - ev[0]) <= 1e-5 * ev[0]
    e.close(); a.close(); b.close()
