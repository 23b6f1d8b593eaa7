"""Trainer: a Python handle on one cs_trainer (one GPU) -- the per-batch work of
`autoencoder.fit` (CAE_improved_modeltrain.py:286-293).  All arithmetic runs in
libcellscreen.so; this file only marshals buffers and splits the flat parameter vector."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib as L
from . import spec
from .engine import _fill_cae
from .spec import CAEWeights


def param_layout(channels=spec.CHANNELS):
    """[(name, shape)] of the flat trainable vector, Keras layer order."""
    out = []
    cin = 1
    for l, cout in enumerate(channels):
        out += [(f"conv{l}.kernel", (3, 3, cin, cout)), (f"conv{l}.bias", (cout,))]
        if l < len(channels) - 1:
            out += [(f"bn{l}.gamma", (cout,)), (f"bn{l}.beta", (cout,))]
        cin = cout
    return out


def moving_layout(channels=spec.CHANNELS):
    out = []
    for l, cout in enumerate(channels[:-1]):
        out += [(f"bn{l}.mean", (cout,)), (f"bn{l}.var", (cout,))]
    return out


def split_flat(flat: np.ndarray, layout):
    out, o = {}, 0
    for name, shape in layout:
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].reshape(shape).copy()
        o += n
    assert o == flat.size
    return out


def flat_from_weights(w: CAEWeights) -> Tuple[np.ndarray, np.ndarray]:
    p, m = [], []
    for l in range(w.n_conv):
        p += [w.kernels[l].ravel(), w.biases[l].ravel()]
        if l < w.n_conv - 1:
            p += [w.bn_gamma[l], w.bn_beta[l]]
            m += [w.bn_mean[l], w.bn_var[l]]
    return np.concatenate(p).astype(np.float32), np.concatenate(m).astype(np.float32)


def weights_from_flat(params: np.ndarray, moving: np.ndarray, bn_eps=spec.BN_EPS, channels=spec.CHANNELS,
                      input_hw=spec.INPUT_HW, n_enc=spec.N_ENC) -> CAEWeights:
    p = split_flat(params, param_layout(channels))
    m = split_flat(moving, moving_layout(channels))
    n = len(channels)
    return CAEWeights([p[f"conv{l}.kernel"] for l in range(n)], [p[f"conv{l}.bias"] for l in range(n)],
                      [p[f"bn{l}.gamma"] for l in range(n - 1)], [p[f"bn{l}.beta"] for l in range(n - 1)],
                      [m[f"bn{l}.mean"] for l in range(n - 1)], [m[f"bn{l}.var"] for l in range(n - 1)],
                      tuple(input_hw), n_enc, bn_eps).validate()


class Trainer:
    def __init__(self, init: CAEWeights, device_id: int = 0, beta1=spec.ADAM_B1, beta2=spec.ADAM_B2,
                 adam_eps=spec.ADAM_EPS, bn_momentum=spec.BN_MOMENTUM):
        self._lib = L.load_library()
        keep: list = []
        w = _fill_cae(init, keep)
        cfg = L.CSTrainCfg(beta1, beta2, adam_eps, bn_momentum, init.bn_eps)
        h = C.c_void_p()
        L.check(self._lib.cs_train_create(C.byref(w), C.byref(cfg), device_id, C.byref(h)))
        self._h = h
        self._device_id = device_id
        self._sync = None
        self._sync_error = None           # the exception a sync-BN all-gather hook raised (it cannot unwind through the C frames)
        self.bn_eps = init.bn_eps
        # the architecture: the reference graph, or any other instance of its layer grammar (csrc/train_generic.hip)
        self.channels, self.input_hw, self.n_enc = tuple(init.channels), tuple(init.input_hw), init.n_enc
        nt, nm = C.c_int64(), C.c_int64()
        L.check(self._lib.cs_train_param_count_of(self._h, C.byref(nt), C.byref(nm)))
        self.n_trainable, self.n_moving = nt.value, nm.value
        assert self.n_trainable == sum(int(np.prod(s)) for _, s in param_layout(self.channels))
        self._grad_tensor = None
        self._fit_train = None            # the training set fit_step was last ordered after

    def close(self):
        if self._h:
            self._lib.cs_train_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buf(self, a):
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a, dtype=np.float32)
            return a, a.ctypes.data, L.CS_MEM_HOST, a.shape[0]
        if not a.is_contiguous():
            a = a.contiguous()
        L.order_after_torch(self._lib.cs_train_wait_stream, self._h, a)    # the batch may still be being gathered on torch's stream
        return a, a.data_ptr(), L.mem_kind(a), a.shape[0]

    def step(self, x, y, lr: float = spec.ADAM_LR) -> Tuple[float, float]:
        """forward + backward + Adam on one batch; returns (loss, mae) of the batch."""
        xb, xp, kind, n = self._buf(x)
        yb, yp, kind2, n2 = self._buf(y)
        assert kind == kind2 and n == n2
        loss, mae = C.c_float(), C.c_float()
        L.check(self._lib.cs_train_step(self._h, xp, yp, n, kind, lr, C.byref(loss), C.byref(mae)))
        return loss.value, mae.value

    def step_async(self, x, y, lr: float = spec.ADAM_LR) -> None:
        """cs_train_step_async: the same batch enqueued without a host synchronisation; its loss / mae go into the running
        sums read_metrics() returns (Keras's epoch metrics: the mean over the epoch's batches)."""
        xb, xp, kind, n = self._buf(x)
        yb, yp, kind2, n2 = self._buf(y)
        assert kind == kind2 and n == n2
        L.check(self._lib.cs_train_step_async(self._h, xp, yp, n, kind, lr))
        L.order_torch_after(self._lib.cs_train_inputs_consumed, self._h, xb, yb)    # torch may reuse the batch's memory only after the copies

    def fit_step(self, train, idx, aug_config=None, seed: int = 0, step: int = 0, lr: float = spec.ADAM_LR) -> None:
        """cs_train_fit_step: ONE library call per fit() batch -- gathers train[idx] (a torch CUDA tensor [n, H, W] resident on this
        trainer's device; idx: a C-contiguous int32 numpy array), draws and applies the batch's augmentation to the input
        (aug_config: ImageDataGenerator.config(), or None), forward + backward + Adam, no host synchronisation.  The caller keeps
        `train` alive and unchanged while steps are in flight; metrics as for step_async."""
        if idx.dtype != np.int32 or not idx.flags["C_CONTIGUOUS"]:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
        if self._fit_train is not train:                      # ordered after whatever produced the training set, once
            if not train.is_contiguous() or str(train.dtype) != "torch.float32":
                raise ValueError("train must be a contiguous float32 CUDA tensor")
            L.order_after_torch(self._lib.cs_train_wait_stream, self._h, train)
            self._fit_train = train
        L.check(self._lib.cs_train_fit_step(self._h, train.data_ptr(), train.shape[0], idx.ctypes.data, idx.shape[0],
                                            C.byref(aug_config) if aug_config is not None else None, int(seed), int(step), lr))

    def read_metrics(self, reset: bool = True) -> Tuple[float, float, int]:
        """(mean loss, mean mae, steps) over the step_async calls since the last reset: one host round trip."""
        lo, ma, st = C.c_double(), C.c_double(), C.c_int64()
        L.check(self._lib.cs_train_read_metrics(self._h, C.byref(lo), C.byref(ma), C.byref(st), 1 if reset else 0))
        return lo.value, ma.value, st.value

    def reset_metrics(self) -> None:
        self.read_metrics(reset=True)

    def forward_backward(self, x, y) -> Tuple[float, float]:
        xb, xp, kind, n = self._buf(x)
        yb, yp, kind2, n2 = self._buf(y)
        assert kind == kind2 and n == n2
        loss, mae = C.c_float(), C.c_float()
        try:
            L.check(self._lib.cs_train_forward_backward(self._h, xp, yp, n, kind, C.byref(loss), C.byref(mae)))
        except L.CellScreenError as e:
            # a failed all-gather hook leaves the peers blocked in their collective: the caller must abort the process group
            if self._sync_error is not None:
                cause, self._sync_error = self._sync_error, None
                raise e from cause
            raise
        return loss.value, mae.value

    def augment(self, x, transforms):
        """cs_train_augment: x [n,64,64] (numpy or torch CUDA), transforms = (CSAugAffine * n) built by
        cellscreen.augment.ImageDataGenerator.pack.  Returns a new array / tensor of the same kind."""
        xb, xp, kind, n = self._buf(x)
        assert len(transforms) == n
        if kind == L.CS_MEM_HOST:
            out = np.empty_like(xb)
            op = out.ctypes.data
        else:
            import torch
            out = torch.empty_like(xb)
            op = out.data_ptr()
            L.order_after_torch(self._lib.cs_train_wait_stream, self._h, out)
        tp = C.c_void_p(transforms.ctypes.data) if isinstance(transforms, np.ndarray) else C.cast(transforms, C.c_void_p)
        L.check(self._lib.cs_train_augment(self._h, xp, n, tp, op, kind))
        L.order_torch_after(self._lib.cs_train_inputs_consumed, self._h, out if kind != L.CS_MEM_HOST else None)   # device output: torch reads it after the kernel
        return out

    def apply(self, lr: float = spec.ADAM_LR):
        """Adam update from the gradient buffer.  With use_grad_tensor() that buffer is a torch tensor an all-reduce may
        still be writing (torch orders its current stream after the collective): the update is ordered after it."""
        L.order_after_torch(self._lib.cs_train_wait_stream, self._h, self._grad_tensor)
        L.check(self._lib.cs_train_apply(self._h, lr))

    def set_sync_bn(self, all_gather, rank: int, world: int):
        """cs_train_set_sync_bn: BatchNormalization statistics over the whole batch when it is split over `world` ranks.
        all_gather(buf, floats_per_rank) must fill buf[0 : world * floats_per_rank] (a torch CUDA float32 tensor in which
        this rank's slot [rank * fpr, (rank + 1) * fpr) is already written) and return when that is complete on the device;
        None switches the synchronisation off."""
        if all_gather is None:
            L.check(self._lib.cs_train_set_sync_bn(self._h, None, None, None, 0, 0, 1))
            self._sync = None
            return
        import torch
        buf = torch.zeros(world * 3 * 64, dtype=torch.float32, device=torch.device("cuda", self._device_id))

        def hook(_ctx, fpr):
            try:
                all_gather(buf, int(fpr))
                return 0
            except Exception as e:  # noqa: BLE001 - an exception must not unwind through the C frames
                self._sync_error = e
                return 1
        cb = L.ALLGATHER_FN(hook)
        L.check(self._lib.cs_train_set_sync_bn(self._h, C.cast(cb, C.c_void_p), None, buf.data_ptr(), buf.numel(), rank, world))
        self._sync = (cb, buf)                # keep the callback and the buffer alive as long as the library may call / write them

    def enable_sync_bn(self, dist, rank: int, world: int, group=None):
        """The same over torch.distributed (backend nccl = RCCL over xGMI): one all_gather_into_tensor per sync point."""
        import torch

        def all_gather(buf, fpr):
            mine = buf[rank * fpr:(rank + 1) * fpr].clone()
            dist.all_gather_into_tensor(buf[:world * fpr], mine, group=group)
            torch.cuda.current_stream(buf.device).synchronize()
        self.set_sync_bn(all_gather, rank, world)

    def use_grad_tensor(self, t):
        """Gradients are written into this torch CUDA float32 tensor (n_trainable elements), so
        torch.distributed can all-reduce it between forward_backward() and apply()."""
        assert t.numel() == self.n_trainable and t.is_cuda and t.is_contiguous()
        self._grad_tensor = t
        L.check(self._lib.cs_train_set_grad_buffer(self._h, t.data_ptr()))

    def evaluate(self, x, y) -> Tuple[float, float]:
        xb, xp, kind, n = self._buf(x)
        yb, yp, kind2, n2 = self._buf(y)
        assert kind == kind2 and n == n2
        loss, mae = C.c_float(), C.c_float()
        L.check(self._lib.cs_train_eval(self._h, xp, yp, n, kind, C.byref(loss), C.byref(mae)))
        return loss.value, mae.value

    def tensor(self, which: int, layer: int, batch: int) -> np.ndarray:
        """Stage tap (parity tests): which 0 relu out, 1 BN out, 2 dz, 3 dBN-out, 4 sigmoid out."""
        rows = spec.layer_table(self.input_hw, self.channels, self.n_enc)
        if which == 4 or (which == 2 and layer == len(self.channels) - 1):
            shape = (batch,) + self.input_hw
        elif which in (0, 2):
            shape = (batch,) + rows[layer]["conv_hw"] + (rows[layer]["cout"],)
        else:
            shape = (batch,) + rows[layer]["out_hw"] + (rows[layer]["cout"],)
        out = np.empty(shape, np.float32)
        L.check(self._lib.cs_train_tensor(self._h, which, layer, batch, out.ctypes.data))
        return out

    def export_flat(self, grads: bool = False):
        p = np.empty(self.n_trainable, np.float32)
        m = np.empty(self.n_moving, np.float32)
        g = np.empty(self.n_trainable, np.float32) if grads else None
        L.check(self._lib.cs_train_export(self._h, p.ctypes.data, m.ctypes.data, g.ctypes.data if grads else None))
        return (p, m, g) if grads else (p, m)

    def weights(self) -> CAEWeights:
        p, m = self.export_flat()
        return weights_from_flat(p, m, self.bn_eps, self.channels, self.input_hw, self.n_enc)

    def load_flat(self, params: Optional[np.ndarray], moving: Optional[np.ndarray]):
        pp = np.ascontiguousarray(params, np.float32) if params is not None else None
        mm = np.ascontiguousarray(moving, np.float32) if moving is not None else None
        L.check(self._lib.cs_train_import(self._h, pp.ctypes.data if pp is not None else None,
                                          mm.ctypes.data if mm is not None else None))
