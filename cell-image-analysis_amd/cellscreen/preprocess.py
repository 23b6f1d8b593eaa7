"""Crop preprocess on the GPU: the two lines the reference runs on every bounding-box crop,

    cell_image_eq      = exposure.equalize_adapthist(cell_image, clip_limit=0.02)
    cell_image_resized = resize(cell_image_eq, (64, 64), anti_aliasing=True)

(improved_detection.py:98-99, CAE_improved_modeltrain.py:92-93), for a whole list of ragged crops
in one call.  All arithmetic happens in libcellscreen.so (csrc/preprocess.hip); there is no CPU
fallback.  The result is the float32 [n,64,64] array `compute_anomaly_scores` builds at
improved_detection.py:122, either as a numpy array or left on the device as a torch tensor."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L

CLIP_LIMIT = 0.02           # improved_detection.py:98
OUT_SIDE = 64               # improved_detection.py:99
PIX_U8, PIX_U16 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1


def pack_crops(crops: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Ragged layout of the C ABI: (pixels 1-D, offsets int64, heights int32, widths int32).
    All crops must share one dtype, uint8 or uint16 (what tifffile returns for a channel)."""
    if len(crops) == 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    dt = np.asarray(crops[0]).dtype
    if dt not in (np.dtype(np.uint8), np.dtype(np.uint16)):
        raise TypeError(f"crop dtype {dt}: the preprocess takes uint8 or uint16 pixels")
    hs = np.empty(len(crops), np.int32)
    ws = np.empty(len(crops), np.int32)
    for i, c in enumerate(crops):
        c = np.asarray(c)
        if c.ndim != 2:
            raise ValueError(f"crop {i} has shape {c.shape}; expected a 2-D bounding-box crop")
        if c.dtype != dt:
            raise TypeError(f"crop {i} is {c.dtype}, crop 0 is {dt}: mixed dtypes")
        hs[i], ws[i] = c.shape
    sizes = hs.astype(np.int64) * ws.astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pixels = np.empty(int(sizes.sum()), dt)
    for i, c in enumerate(crops):
        pixels[offsets[i]:offsets[i] + sizes[i]] = np.asarray(c).ravel()
    return pixels, offsets, hs, ws


class Preprocessor:
    """One cs_preproc handle (one GPU, one stream)."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load_library()
        self._h = C.c_void_p()
        L.check(self._lib.cs_preproc_create(device_id, C.byref(self._h)))
        self.device_id = device_id

    def close(self):
        if self._h:
            self._lib.cs_preproc_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_packed(self, pixels, offsets: np.ndarray, heights: np.ndarray, widths: np.ndarray,
                   clip_limit: float = CLIP_LIMIT, out=None, want_clahe: bool = False):
        """pixels: 1-D numpy array (host) or torch CUDA tensor (device) of uint8/uint16.
        out: None (numpy result), or a torch CUDA float32 tensor [n,64,64] to fill in place.
        Returns out, or (out, clahe_u16) with want_clahe."""
        n = int(len(offsets))
        offsets = np.ascontiguousarray(offsets, np.int64)
        heights = np.ascontiguousarray(heights, np.int32)
        widths = np.ascontiguousarray(widths, np.int32)
        on_dev = not isinstance(pixels, np.ndarray)
        if on_dev:
            import torch
            if pixels.dtype == torch.uint8:
                ptype = PIX_U8
            elif pixels.dtype in (torch.uint16, torch.int16):
                ptype = PIX_U16
            else:
                raise TypeError(f"pixel tensor dtype {pixels.dtype}")
            if not pixels.is_cuda or not pixels.is_contiguous():
                raise ValueError("device pixels must be a contiguous CUDA tensor")
            pix_ptr, n_pix = pixels.data_ptr(), pixels.numel()
        else:
            pixels = np.ascontiguousarray(pixels)
            if pixels.dtype == np.uint8:
                ptype = PIX_U8
            elif pixels.dtype == np.uint16:
                ptype = PIX_U16
            else:
                raise TypeError(f"pixel dtype {pixels.dtype}: uint8 or uint16 expected")
            pix_ptr, n_pix = pixels.ctypes.data, pixels.size
        clahe = None
        if out is None:
            res = np.empty((n, OUT_SIDE, OUT_SIDE), np.float32)
            out_ptr, out_kind = res.ctypes.data, MEM_HOST
            if want_clahe:
                clahe = np.zeros(n_pix, np.uint16)
                cl_ptr = clahe.ctypes.data
        else:
            import torch
            if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                    and tuple(out.shape) == (n, OUT_SIDE, OUT_SIDE)):
                raise ValueError("out must be a contiguous CUDA float32 tensor of shape [n,64,64]")
            res, out_ptr, out_kind = out, out.data_ptr(), MEM_DEVICE
            if want_clahe:
                clahe = torch.zeros(n_pix, dtype=torch.int16, device=out.device)     # gaps between crops read 0; ordered before the kernel below
                cl_ptr = clahe.data_ptr()
        L.order_after_torch(self._lib.cs_preproc_wait_stream, self._h, pixels if on_dev else None, out, clahe if out is not None else None)
        L.check(self._lib.cs_preprocess(self._h, pix_ptr, ptype, n_pix, MEM_DEVICE if on_dev else MEM_HOST,
                                        offsets.ctypes.data, heights.ctypes.data, widths.ctypes.data, n,
                                        float(clip_limit), out_ptr, cl_ptr if want_clahe else None, out_kind))
        return (res, clahe) if want_clahe else res

    def __call__(self, crops: Sequence[np.ndarray], clip_limit: float = CLIP_LIMIT) -> np.ndarray:
        """List of 2-D uint8/uint16 crops -> float32 [n,64,64]."""
        pixels, offsets, hs, ws = pack_crops(crops)
        return self.run_packed(pixels, offsets, hs, ws, clip_limit)

    def last_timing(self) -> Tuple[float, int]:
        ms, px = C.c_double(), C.c_int64()
        L.check(self._lib.cs_preproc_last_timing(self._h, C.byref(ms), C.byref(px)))
        return ms.value, px.value


def split_clahe(clahe: np.ndarray, offsets, heights, widths) -> List[np.ndarray]:
    """Ragged stage tap -> list of uint16 [H,W] images."""
    return [np.asarray(clahe[o:o + h * w]).reshape(h, w) for o, h, w in zip(offsets, heights, widths)]
