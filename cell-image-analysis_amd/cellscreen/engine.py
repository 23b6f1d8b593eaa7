"""Engine: a Python handle on one cs_model (one GPU, one stream, one workspace).
Marshals numpy arrays / torch CUDA tensors into the C ABI; all arithmetic happens in
libcellscreen.so.  Buffers may be numpy arrays (host) or torch CUDA tensors (device)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib as L
from . import spec
from .spec import CAEWeights, DetectorParams


def _fill_cae(w: CAEWeights, keep: list) -> L.CSCaeWeights:
    s = L.CSCaeWeights()
    s.height, s.width = w.input_hw
    s.n_conv, s.n_enc = w.n_conv, w.n_enc
    s.bn_eps = w.bn_eps

    def put(field, l, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        keep.append(a)
        getattr(s, field)[l] = a.ctypes.data

    for l in range(w.n_conv):
        s.channels[l] = w.channels[l]
        put("kernel", l, w.kernels[l])
        put("bias", l, w.biases[l])
    for l in range(len(w.bn_gamma)):
        put("bn_gamma", l, w.bn_gamma[l]); put("bn_beta", l, w.bn_beta[l])
        put("bn_mean", l, w.bn_mean[l]); put("bn_var", l, w.bn_var[l])
    return s


def _fill_det(d: DetectorParams, keep: list) -> L.CSDetectorParams:
    s = L.CSDetectorParams()
    s.n_features, s.n_components = d.n_features, d.n_components

    def addr(arr, dt):
        a = np.ascontiguousarray(arr, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    s.scaler_center = addr(d.scaler_center, np.float32)
    s.scaler_scale = addr(d.scaler_scale, np.float64)
    s.pca_components = addr(d.pca_components, np.float32)
    s.pca_mean_proj = addr(d.pca_mean_proj, np.float32)
    for name in ("conservative", "moderate"):
        p = getattr(d, name)
        o = getattr(s, name)
        o.n_sv = p.n_sv
        o.support_vectors = addr(p.support_vectors, np.float64)
        o.dual_coef = addr(np.ravel(p.dual_coef), np.float64)
        o.gamma, o.rho = float(p.gamma), float(p.rho)
    return s


class Engine:
    def __init__(self, handle):
        self._h = handle
        self._lib = L.load_library()
        info = L.CSModelInfo()
        L.check(self._lib.cs_model_get_info(self._h, C.byref(info)))
        self.info = info
        self.precision = "fp32_exact" if info.precision == L.PRECISION_FP32_EXACT else "split16"

    # ---- construction -------------------------------------------------------------
    @classmethod
    def from_weights(cls, autoencoder: CAEWeights, encoder: Optional[CAEWeights] = None,
                     detector: Optional[DetectorParams] = None, device_id: int = 0, precision="split16",
                     debug_flags: int = 0) -> "Engine":
        """precision: "split16" (default: fp32 contractions as two-term fp16 splits on the 16-bit matrix instructions, inside
        every fp32 tolerance) or "fp32_exact" (every contraction on the fp32 matrix instructions: the reference's own
        arithmetic, improved_detection.py:122,125,130).  debug_flags: L.DEBUG_* (unfused forms, for A/B runs and tests)."""
        lib = L.load_library()
        opts = L.model_options(precision, debug_flags)
        keep: list = []
        ae = _fill_cae(autoencoder, keep)
        en = _fill_cae(encoder, keep) if encoder is not None else None
        de = _fill_det(detector, keep) if detector is not None else None
        h = C.c_void_p()
        L.check(lib.cs_model_from_arrays(C.byref(ae), C.byref(en) if en is not None else None,
                                         C.byref(de) if de is not None else None, device_id, C.byref(opts), C.byref(h)))
        return cls(h)

    @classmethod
    def from_model_dir(cls, model_dir: str, device_id: int = 0, precision="split16", debug_flags: int = 0) -> "Engine":
        """model_dir: the native file set, or the reference's six files (improved_detection.py:28-41: two `.keras`
        archives + four pickles), which are converted on the spot.  precision: see from_weights."""
        from . import model_io
        model_dir = model_io.ensure_native_model_dir(model_dir)
        lib = L.load_library()
        h = C.c_void_p()
        opts = L.model_options(precision, debug_flags)
        L.check(lib.cs_model_load(model_dir.encode(), device_id, C.byref(opts), C.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            self._lib.cs_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chunk(self, cells: int):
        L.check(self._lib.cs_model_set_chunk(self._h, int(cells)))
        self.info.chunk_cells = int(cells)

    # ---- helpers --------------------------------------------------------------------
    def _crops(self, crops):
        """-> (buffer, n, kind); numpy input is made float32 C-contiguous (N,H,W)."""
        if isinstance(crops, np.ndarray) or not hasattr(crops, "data_ptr"):
            a = np.ascontiguousarray(crops, dtype=np.float32)
            if a.ndim == 4 and a.shape[-1] == 1:
                a = a[..., 0]
            if a.ndim != 3 or a.shape[1:] != (self.info.height, self.info.width):
                raise ValueError(f"crops must be (N,{self.info.height},{self.info.width}), got {a.shape}")
            return a, a.shape[0], L.CS_MEM_HOST
        t = crops
        if str(t.dtype) != "torch.float32" or not t.is_contiguous():
            raise ValueError("device crops must be a contiguous float32 tensor")
        if t.dim() == 4 and t.shape[-1] == 1:
            t = t[..., 0]
        if t.dim() != 3 or tuple(t.shape[1:]) != (self.info.height, self.info.width):
            raise ValueError(f"crops must be (N,{self.info.height},{self.info.width}), got {tuple(t.shape)}")
        L.order_after_torch(self._lib.cs_model_wait_stream, self._h, t)
        return t, t.shape[0], L.mem_kind(t)

    @staticmethod
    def _alloc(like_kind, like, shape, dtype):
        if like_kind == L.CS_MEM_HOST:
            return np.empty(shape, dtype=dtype)
        import torch
        tdt = {np.float32: torch.float32, np.float64: torch.float64, np.int8: torch.int8}[dtype]
        return torch.empty(shape, dtype=tdt, device=like.device)

    # ---- the hot path ---------------------------------------------------------------
    def screen(self, crops, out: Optional[Dict] = None, out_device: Optional[bool] = None) -> Dict:
        """cs_screen.  Returns dict(mse, mae, cons_score, mod_score, cons_pred, mod_pred);
        arrays live where the crops live unless out_device says otherwise."""
        buf, n, kind = self._crops(crops)
        okind = kind if out_device is None else (L.CS_MEM_DEVICE if out_device else L.CS_MEM_HOST)
        if out is None:
            like = buf if okind == L.CS_MEM_DEVICE else None
            out = dict(mse=self._alloc(okind, like, (n,), np.float32), mae=self._alloc(okind, like, (n,), np.float32),
                       cons_score=self._alloc(okind, like, (n,), np.float64), mod_score=self._alloc(okind, like, (n,), np.float64),
                       cons_pred=self._alloc(okind, like, (n,), np.int8), mod_pred=self._alloc(okind, like, (n,), np.int8))
        L.check(self._lib.cs_screen(self._h, L._ptr(buf), n, kind, L._ptr(out["mse"]), L._ptr(out["mae"]),
                                    L._ptr(out["cons_score"]), L._ptr(out["mod_score"]),
                                    L._ptr(out["cons_pred"]), L._ptr(out["mod_pred"]), okind))
        return out

    def reconstruct(self, crops, want_recon: bool = True):
        buf, n, kind = self._crops(crops)
        rec = self._alloc(kind, buf, (n, self.info.height, self.info.width), np.float32) if want_recon else None
        mse = self._alloc(kind, buf, (n,), np.float32)
        mae = self._alloc(kind, buf, (n,), np.float32)
        L.check(self._lib.cs_reconstruct(self._h, L._ptr(buf), n, kind, L._ptr(rec), L._ptr(mse), L._ptr(mae), kind))
        return rec, mse, mae

    def encode(self, crops, which: int = 1):
        buf, n, kind = self._crops(crops)
        f = self._alloc(kind, buf, (n, self.info.feature_dim), np.float32)
        L.check(self._lib.cs_encode(self._h, L._ptr(buf), n, kind, which, L._ptr(f), kind))
        return f

    def layer_output(self, crops, layer: int):
        buf, n, kind = self._crops(crops)
        rows = spec.layer_table((self.info.height, self.info.width), tuple(self.info.channels[:self.info.n_conv]), self.info.n_enc)
        oh, ow = rows[layer]["out_hw"]
        o = self._alloc(kind, buf, (n, oh, ow, rows[layer]["cout"]), np.float32)
        L.check(self._lib.cs_layer_output(self._h, L._ptr(buf), n, kind, layer, L._ptr(o), kind))
        return o

    def scaler_pca(self, features: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(features, dtype=np.float32)
        o = np.empty((f.shape[0], self.info.n_components), dtype=np.float32)
        L.check(self._lib.cs_scaler_pca(self._h, f.ctypes.data, f.shape[0], L.CS_MEM_HOST, o.ctypes.data, L.CS_MEM_HOST))
        return o

    def svm_decision(self, pca: np.ndarray):
        p = np.ascontiguousarray(pca, dtype=np.float32)
        c = np.empty(p.shape[0], dtype=np.float64)
        m = np.empty(p.shape[0], dtype=np.float64)
        L.check(self._lib.cs_svm_decision(self._h, p.ctypes.data, p.shape[0], L.CS_MEM_HOST, c.ctypes.data, m.ctypes.data, L.CS_MEM_HOST))
        return c, m

    def synth_crops(self, seed: int, first_cell: int, out_tensor):
        """Fills a torch CUDA tensor (n,H,W) float32 on this engine's device."""
        n = out_tensor.shape[0]
        npix = int(np.prod(out_tensor.shape[1:]))
        L.order_after_torch(self._lib.cs_model_wait_stream, self._h, out_tensor)     # e.g. after a torch.empty + earlier users
        L.check(self._lib.cs_synth_crops(self._h, seed, first_cell, n, npix, out_tensor.data_ptr()))
        return out_tensor

    # ---- measurement ------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        L.check(self._lib.cs_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        L.check(self._lib.cs_profile_reset(self._h))

    def profile(self) -> Dict[str, dict]:
        out = {}
        for k in range(self._lib.cs_profile_kernel_count()):
            ms, ln, cells, fl = C.c_double(), C.c_int64(), C.c_int64(), C.c_double()
            L.check(self._lib.cs_profile_get(self._h, k, C.byref(ms), C.byref(ln), C.byref(cells), C.byref(fl)))
            mf = C.c_double()
            L.check(self._lib.cs_profile_mfma_per_cell(self._h, k, C.byref(mf)))
            bf = C.c_double()
            L.check(self._lib.cs_profile_bf16_mfma_per_cell(self._h, k, C.byref(bf)))
            out[self._lib.cs_profile_kernel_name(k).decode()] = dict(ms=ms.value, launches=ln.value, cells=cells.value, flops=fl.value,
                                                                      mfma_per_cell=mf.value, bf16_mfma_per_cell=bf.value)
        return out
