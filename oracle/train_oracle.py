"""CPU oracle of the CAE training step -- TEST INFRASTRUCTURE ONLY (see cae_oracle.c header).

Restates, in numpy float64 (or float32), one `autoencoder.fit` step of
CAE_improved_modeltrain.py:286-293 on the graph of :184-229 compiled at :223-227:
  forward with BatchNormalization in training mode (batch mean / biased variance, Keras
  momentum 0.99 update of the moving statistics), loss = 'mse' (mean over every element),
  metric 'mae', backward, Adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7).
Input and target are separate arrays because the reference trains on
`datagen.flow(X_train, X_train)` (:287): augmented input, un-augmented target.

The arithmetic lives in TensorFlow/Keras (absent here, unpinned by the reference): PARITY
UNPINNED by the reference itself; pinned in tests/test_train_oracle_cpu.py against torch
autograd + torch.optim.Adam (an independent implementation) on the same graph.

Moving-variance convention: Keras 3 updates moving_variance with the biased batch variance
(`ops.moments`); TF2's fused kernel used the unbiased one.  `.keras` files are Keras 3, so
biased is used here and in the HIP trainer; the difference is a factor N/(N-1), N = B*H*W >= 2048.
"""
from __future__ import annotations

import numpy as np


def _im2col(x, ups):
    """x: (N,h,w,C) stored input; returns (N,H,W,9*C) patches of the (optionally x2 nearest
    upsampled) zero-padded input, tap-major (tap = (dy+1)*3 + (dx+1)) then channel."""
    if ups:
        x = x.repeat(2, axis=1).repeat(2, axis=2)           # UpSampling2D((2,2)), nearest
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C), dtype=x.dtype)
    xp[:, 1:-1, 1:-1] = x
    cols = [xp[:, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)]
    return np.concatenate(cols, axis=-1)


def _col2im(dcols, C, ups):
    """adjoint of _im2col: (N,H,W,9*C) -> gradient wrt the stored input."""
    N, H, W, _ = dcols.shape
    dxp = np.zeros((N, H + 2, W + 2, C), dtype=dcols.dtype)
    t = 0
    for dy in range(3):
        for dx in range(3):
            dxp[:, dy:dy + H, dx:dx + W] += dcols[..., t * C:(t + 1) * C]
            t += 1
    dx_ = dxp[:, 1:-1, 1:-1]
    if ups:                                                  # adjoint of nearest x2: 2x2 sum
        dx_ = dx_.reshape(N, H // 2, 2, W // 2, 2, C).sum(axis=(2, 4))
    return dx_


class TrainState:
    """Trainable parameters + BN moving statistics + Adam slots, as lists per conv layer."""

    def __init__(self, w, dtype=np.float64):
        self.dtype = dtype
        self.n_conv, self.n_enc, self.bn_eps = w.n_conv, w.n_enc, float(w.bn_eps)
        f = lambda a: np.array(a, dtype=dtype)
        self.kernels = [f(k) for k in w.kernels]
        self.biases = [f(b) for b in w.biases]
        self.gamma = [f(g) for g in w.bn_gamma]
        self.beta = [f(b) for b in w.bn_beta]
        self.mov_mean = [f(m) for m in w.bn_mean]
        self.mov_var = [f(v) for v in w.bn_var]
        self.step = 0
        self.m = [np.zeros_like(p) for p in self.trainables()]
        self.v = [np.zeros_like(p) for p in self.trainables()]

    def trainables(self):
        """Order: conv0.kernel, conv0.bias, bn0.gamma, bn0.beta, conv1.kernel, ... (Keras layer order)."""
        out = []
        for l in range(self.n_conv):
            out += [self.kernels[l], self.biases[l]]
            if l < self.n_conv - 1:
                out += [self.gamma[l], self.beta[l]]
        return out


def forward_backward(st: TrainState, x, y, momentum=0.99, update_moving=True, relu_masks=None, pool_args=None):
    """One training-mode forward + backward.  x (aug input), y (target): (N,H,W).
    Returns dict(loss, mae, out, grads (same order as st.trainables()), batch stats).

    relu_masks / pool_args (lists per layer, entries may be None) override the two places where
    the gradient is discontinuous -- ReLU's derivative at z = 0 and MaxPooling2D's routing at a
    tie -- with another evaluation's decisions (r > 0 masks; arg-max index 0..3 in (dy,dx) order).
    An fp32 and an fp64 evaluation disagree on a handful of the ~1e6 such decisions per layer
    (|z| within rounding of 0); gradient parity is only meaningful on the same activation pattern."""
    dt = st.dtype
    x = np.asarray(x, dtype=dt)[..., None]
    y = np.asarray(y, dtype=dt)[..., None]
    cache = []
    h = x
    batch_mean, batch_var = [], []
    margins = []          # per BN layer: how far the layer's discontinuous decisions are from flipping, relative to its range
    for l in range(st.n_conv):
        ups = l > st.n_enc
        cols = _im2col(h, ups)
        K = st.kernels[l].reshape(-1, st.kernels[l].shape[3])                   # (9*cin, cout)
        z = cols @ K + st.biases[l]
        if l == st.n_conv - 1:
            out = 1.0 / (1.0 + np.exp(-z))                                       # sigmoid
            cache.append(dict(cols=cols, ups=ups, cin=h.shape[3]))
            break
        r = np.maximum(z, 0)                                                     # relu inside Conv2D
        mu = r.mean(axis=(0, 1, 2))
        var = r.var(axis=(0, 1, 2))                                              # biased
        inv = 1.0 / np.sqrt(var + dt(st.bn_eps))
        xhat = (r - mu) * inv
        yb = st.gamma[l] * xhat + st.beta[l]
        batch_mean.append(mu); batch_var.append(var)
        if update_moving:
            st.mov_mean[l] = st.mov_mean[l] * momentum + mu * (1 - momentum)
            st.mov_var[l] = st.mov_var[l] * momentum + var * (1 - momentum)
        c = dict(cols=cols, ups=ups, cin=h.shape[3], r=r, xhat=xhat, inv=inv)
        mg = dict(relu=float(np.abs(z).min() / max(np.abs(z).max(), 1e-300)))   # min |z| over the layer / max |z|
        if l < st.n_enc:
            N, H, W, C = yb.shape
            win = yb.reshape(N, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, C, 4)
            arg = win.argmax(axis=-1)                                            # first max in (dy,dx) order
            # smallest lead of a window's winner over its runner-up / max |BN out|, over the windows where the routing matters
            # (a tie between two elements that ReLU zeroed sends the gradient to an element whose ReLU' is 0 either way)
            order = np.argsort(win, axis=-1)
            rwin = r.reshape(N, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, C, 4)
            v1, v2 = np.take_along_axis(win, order[..., 3:4], -1), np.take_along_axis(win, order[..., 2:3], -1)
            r1, r2 = np.take_along_axis(rwin, order[..., 3:4], -1), np.take_along_axis(rwin, order[..., 2:3], -1)
            live = (r1 > 0) | (r2 > 0)
            mg["pool"] = float((v1 - v2)[live].min() / max(np.abs(yb).max(), 1e-300)) if live.any() else 1.0
            if pool_args is not None and pool_args[l] is not None:
                arg = np.asarray(pool_args[l])
            h = np.take_along_axis(win, arg[..., None], axis=-1)[..., 0]
            c["arg"] = arg
        else:
            h = yb
        margins.append(mg)
        cache.append(c)
    n_el = out.size
    diff = out - y
    loss = float((diff * diff).sum() / n_el)                                     # loss='mse'
    mae = float(np.abs(diff).sum() / n_el)                                       # metrics=['mae']

    grads = [None] * len(st.trainables())
    gi = len(grads)
    dz = (2.0 / n_el) * diff * out * (1.0 - out)
    dzs, das = [None] * st.n_conv, [None] * st.n_conv
    for l in range(st.n_conv - 1, -1, -1):
        c = cache[l]
        if l < st.n_conv - 1:
            # gradient arriving at the BN output (post-pool for the encoder)
            dyb = dh
            if l < st.n_enc:
                N, Ho, Wo, C = dyb.shape
                dwin = np.zeros((N, Ho, Wo, C, 4), dtype=dt)
                np.put_along_axis(dwin, c["arg"][..., None], dyb[..., None], axis=-1)
                dyb = dwin.reshape(N, Ho, Wo, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(N, Ho * 2, Wo * 2, C)
            nred = dyb.shape[0] * dyb.shape[1] * dyb.shape[2]
            dbeta = dyb.sum(axis=(0, 1, 2))
            dgamma = (dyb * c["xhat"]).sum(axis=(0, 1, 2))
            dr = (st.gamma[l] * c["inv"] / nred) * (nred * dyb - dbeta - c["xhat"] * dgamma)
            mask = (c["r"] > 0) if (relu_masks is None or relu_masks[l] is None) else np.asarray(relu_masks[l])
            dz = dr * mask
            gi -= 2
            grads[gi], grads[gi + 1] = dgamma, dbeta
        dzs[l] = dz
        K = st.kernels[l].reshape(-1, st.kernels[l].shape[3])
        gi -= 2
        grads[gi] = (c["cols"].reshape(-1, K.shape[0]).T @ dz.reshape(-1, K.shape[1])).reshape(st.kernels[l].shape)
        grads[gi + 1] = dz.sum(axis=(0, 1, 2))
        if l > 0:
            dh = _col2im(dz @ K.T, c["cin"], c["ups"])
            das[l - 1] = dh
    return dict(loss=loss, mae=mae, out=out[..., 0], grads=grads, batch_mean=batch_mean, batch_var=batch_var,
                dz=dzs, da=das, relu=[c.get("r") for c in cache], margins=margins)


def adam_step(st: TrainState, grads, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; w -= alpha*m/(sqrt(v)+eps)."""
    st.step += 1
    t = st.step
    dt = st.dtype
    alpha = dt(lr) * np.sqrt(dt(1.0) - dt(b2) ** t) / (dt(1.0) - dt(b1) ** t)
    for p, g, m, v in zip(st.trainables(), grads, st.m, st.v):
        m += (g - m) * dt(1 - b1)
        v += (g * g - v) * dt(1 - b2)
        p -= (m * alpha) / (np.sqrt(v) + dt(eps))


def train_step(st: TrainState, x, y, lr=1e-3):
    r = forward_backward(st, x, y)
    adam_step(st, r["grads"], lr=lr)
    return r


def evaluate(st: TrainState, x, y):
    """Inference-mode loss / mae with the moving statistics (validation pass of fit())."""
    dt = st.dtype
    h = np.asarray(x, dtype=dt)[..., None]
    y = np.asarray(y, dtype=dt)[..., None]
    for l in range(st.n_conv):
        cols = _im2col(h, l > st.n_enc)
        z = cols @ st.kernels[l].reshape(-1, st.kernels[l].shape[3]) + st.biases[l]
        if l == st.n_conv - 1:
            out = 1.0 / (1.0 + np.exp(-z))
            break
        r = np.maximum(z, 0)
        yb = st.gamma[l] * (r - st.mov_mean[l]) / np.sqrt(st.mov_var[l] + dt(st.bn_eps)) + st.beta[l]
        if l < st.n_enc:
            N, H, W, C = yb.shape
            yb = yb.reshape(N, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))
        h = yb
    d = out - y
    return float((d * d).mean()), float(np.abs(d).mean())
