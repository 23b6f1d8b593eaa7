"""Pins oracle/train_oracle.py (numpy restatement of one fit() step,
CAE_improved_modeltrain.py:223-227, 286-293) against torch autograd + torch.optim.Adam."""
import numpy as np
import pytest

import helpers as H  # noqa: F401
from cellscreen import synth
from oracle import train_oracle as T

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


def torch_model(w):
    P = lambda a: torch.nn.Parameter(torch.from_numpy(np.array(a, dtype=np.float64)))
    p = dict(k=[P(k) for k in w.kernels], b=[P(b) for b in w.biases], g=[P(g) for g in w.bn_gamma], be=[P(b) for b in w.bn_beta])
    rm = [torch.from_numpy(np.array(m, dtype=np.float64)) for m in w.bn_mean]
    rv = [torch.from_numpy(np.array(v, dtype=np.float64)) for v in w.bn_var]
    return p, rm, rv


def torch_forward(p, rm, rv, x, w, train=True):
    h = torch.from_numpy(x).double()[:, None]
    for l in range(7):
        if l > 3:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
        h = F.conv2d(h, p["k"][l].permute(3, 2, 0, 1), p["b"][l], padding=1)
        if l < 6:
            h = F.relu(h)
            if train:
                n = h.numel() // h.shape[1]
                mu = h.mean(dim=(0, 2, 3)); var = h.var(dim=(0, 2, 3), unbiased=False)
                with torch.no_grad():   # Keras 3 convention: biased variance in the moving average
                    rm[l].mul_(0.99).add_(mu.detach() * 0.01); rv[l].mul_(0.99).add_(var.detach() * 0.01)
                h = (h - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + w.bn_eps) * p["g"][l][None, :, None, None] + p["be"][l][None, :, None, None]
            else:
                h = F.batch_norm(h, rm[l], rv[l], p["g"][l], p["be"][l], False, 0.0, w.bn_eps)
            if l < 3:
                h = F.max_pool2d(h, 2)
        else:
            h = torch.sigmoid(h)
    return h[:, 0]


def params_list(p):
    out = []
    for l in range(7):
        out += [p["k"][l], p["b"][l]]
        if l < 6:
            out += [p["g"][l], p["be"][l]]
    return out


def test_gradients_and_adam_match_torch():
    w = synth.random_cae(seed=3)
    x = synth.blob_crops(1, 4)
    y = np.clip(x + 0.02 * np.random.default_rng(0).standard_normal(x.shape).astype(np.float32), 0, 1)   # input != target
    st = T.TrainState(w)
    p, rm, rv = torch_model(w)
    for step in range(3):
        r = T.forward_backward(st, x, y)
        out = torch_forward(p, rm, rv, x, w, train=True)
        loss = ((out - torch.from_numpy(y).double()) ** 2).mean()
        for tp in params_list(p):
            tp.grad = None
        loss.backward()
        assert abs(r["loss"] - loss.item()) <= 1e-12 * max(1, abs(loss.item()))
        for g, tp in zip(r["grads"], params_list(p)):
            tg = tp.grad.numpy()
            assert g.shape == tg.shape
            assert np.linalg.norm(g - tg) <= 1e-9 * max(np.linalg.norm(tg), 1e-30), step
        T.adam_step(st, r["grads"])
        # torch.optim.Adam adds eps to the bias-corrected sqrt(v_hat); Keras folds the correction
        # into alpha and adds eps to the raw sqrt(v): not the same update, so the Keras formula is
        # pinned separately (test_adam_formula_is_keras) and the torch parameters just follow ours.
        with torch.no_grad():
            for tp, mine in zip(params_list(p), st.trainables()):
                tp.copy_(torch.from_numpy(mine))
    for l in range(6):
        assert np.allclose(st.mov_mean[l], rm[l].numpy(), rtol=1e-12, atol=1e-14)
        assert np.allclose(st.mov_var[l], rv[l].numpy(), rtol=1e-12, atol=1e-14)


def test_adam_formula_is_keras():
    """alpha = lr*sqrt(1-b2^t)/(1-b1^t); w -= alpha*m/(sqrt(v)+eps)  (Keras), on a scalar."""
    class W:   # minimal stand-in
        n_conv, n_enc, bn_eps = 1, 0, 1e-3
        kernels = [np.ones((3, 3, 1, 1))]; biases = [np.zeros(1)]
        bn_gamma = bn_beta = bn_mean = bn_var = []
    st = T.TrainState(W)
    g = [np.full((3, 3, 1, 1), 0.5), np.full(1, -2.0)]
    m = v = 0.0; wref = 1.0
    for t in range(1, 4):
        T.adam_step(st, g, lr=1e-3)
        m = 0.9 * m + 0.1 * 0.5; v = 0.999 * v + 0.001 * 0.25
        alpha = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        wref -= alpha * m / (np.sqrt(v) + 1e-7)
        assert abs(st.kernels[0][0, 0, 0, 0] - wref) < 1e-15


def test_evaluate_matches_inference_forward():
    from oracle import oracle
    w = synth.random_cae(seed=4)
    x = synth.synth_crops(2, 0, 3)
    st = T.TrainState(w)
    loss, mae = T.evaluate(st, x, x)
    r = oracle.cae_forward(w, x, acc64=True)
    assert abs(loss - r["mse"].mean()) < 1e-6 * loss and abs(mae - r["mae"].mean()) < 1e-6 * mae


def test_loss_decreases_on_blobs():
    w = synth.random_cae(seed=6, trivial_bn=True)
    x = synth.blob_crops(2, 8)
    st = T.TrainState(w, dtype=np.float64)
    l0 = T.train_step(st, x, x)["loss"]
    for _ in range(6):
        l1 = T.train_step(st, x, x)["loss"]
    assert l1 < l0
