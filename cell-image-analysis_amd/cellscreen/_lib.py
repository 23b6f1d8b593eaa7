"""ctypes binding of libcellscreen.so (include/cellscreen.h).  Thin by design: argument
marshalling and error translation only.  No computation happens here and there is no
fallback: if the library or a GPU is missing, calls raise."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

CS_MAX_CONV = 16
CS_MEM_HOST, CS_MEM_DEVICE = 0, 1
_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libcellscreen.so")


class CellScreenError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libcellscreen status {status}: {message}")
        self.status = status


class CSAugAffine(C.Structure):
    _fields_ = [("m", C.c_double * 4), ("off", C.c_double * 2), ("identity", C.c_int32), ("flip_h", C.c_int32),
                ("flip_v", C.c_int32), ("reserved", C.c_int32)]


class CSAugConfig(C.Structure):
    _fields_ = [("rotation_range", C.c_double), ("width_shift_range", C.c_double), ("height_shift_range", C.c_double),
                ("zoom_lo", C.c_double), ("zoom_hi", C.c_double), ("horizontal_flip", C.c_int32), ("vertical_flip", C.c_int32),
                ("center", C.c_double)]


class CSCaeWeights(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("n_conv", C.c_int32), ("n_enc", C.c_int32),
                ("channels", C.c_int32 * CS_MAX_CONV),
                ("kernel", C.c_void_p * CS_MAX_CONV), ("bias", C.c_void_p * CS_MAX_CONV),
                ("bn_gamma", C.c_void_p * CS_MAX_CONV), ("bn_beta", C.c_void_p * CS_MAX_CONV),
                ("bn_mean", C.c_void_p * CS_MAX_CONV), ("bn_var", C.c_void_p * CS_MAX_CONV),
                ("bn_eps", C.c_float)]


class CSOcsvmParams(C.Structure):
    _fields_ = [("n_sv", C.c_int32), ("support_vectors", C.c_void_p), ("dual_coef", C.c_void_p),
                ("gamma", C.c_double), ("rho", C.c_double)]


class CSDetectorParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_components", C.c_int32),
                ("scaler_center", C.c_void_p), ("scaler_scale", C.c_void_p),
                ("pca_components", C.c_void_p), ("pca_mean_proj", C.c_void_p),
                ("conservative", CSOcsvmParams), ("moderate", CSOcsvmParams)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64)      # cs_allgather_fn


class CSTrainCfg(C.Structure):
    _fields_ = [("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float)]


# cs_precision / CS_DEBUG_* of include/cellscreen.h
PRECISION_SPLIT16, PRECISION_FP32_EXACT = 0, 1
PRECISIONS = {"split16": PRECISION_SPLIT16, "fp32_exact": PRECISION_FP32_EXACT, "exact": PRECISION_FP32_EXACT,
              PRECISION_SPLIT16: PRECISION_SPLIT16, PRECISION_FP32_EXACT: PRECISION_FP32_EXACT}
DEBUG_NO_FUSE12, DEBUG_NO_FUSE45, DEBUG_NO_FUSE67, DEBUG_NO_SMALL_SPLIT = 1, 2, 4, 8


class CSModelOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("precision", C.c_int32), ("debug_flags", C.c_uint32), ("reserved", C.c_uint32 * 5)]


def model_options(precision="split16", debug_flags: int = 0) -> "CSModelOptions":
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be 'split16' or 'fp32_exact', got {precision!r}")
    o = CSModelOptions()
    o.struct_size = C.sizeof(CSModelOptions)
    o.precision = PRECISIONS[precision]
    o.debug_flags = int(debug_flags)
    return o


class CSModelInfo(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("n_conv", C.c_int32), ("n_enc", C.c_int32),
                ("feature_dim", C.c_int32), ("n_components", C.c_int32),
                ("n_sv_conservative", C.c_int32), ("n_sv_moderate", C.c_int32),
                ("shared_encoder", C.c_int32), ("has_detector", C.c_int32), ("device_id", C.c_int32),
                ("chunk_cells", C.c_int64), ("channels", C.c_int32 * CS_MAX_CONV), ("reference_arch", C.c_int32),
                ("precision", C.c_int32), ("debug_flags", C.c_uint32)]


# every exported symbol of include/cellscreen.h: (restype, argtypes)
_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
SIGNATURES = {
    "cs_abi_version": (_I, []),
    "cs_status_string": (C.c_char_p, [_I]),
    "cs_last_error": (C.c_char_p, []),
    "cs_device_count": (_I, []),
    "cs_model_load": (_I, [C.c_char_p, _I, C.POINTER(CSModelOptions), C.POINTER(_P)]),
    "cs_model_from_arrays": (_I, [C.POINTER(CSCaeWeights), C.POINTER(CSCaeWeights), C.POINTER(CSDetectorParams), _I,
                                  C.POINTER(CSModelOptions), C.POINTER(_P)]),
    "cs_model_free": (None, [_P]),
    "cs_model_wait_stream": (_I, [_P, _P]),
    "cs_model_get_info": (_I, [_P, C.POINTER(CSModelInfo)]),
    "cs_model_set_chunk": (_I, [_P, _L]),
    "cs_screen": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _I]),
    "cs_reconstruct": (_I, [_P, _P, _L, _I, _P, _P, _P, _I]),
    "cs_encode": (_I, [_P, _P, _L, _I, _I, _P, _I]),
    "cs_layer_output": (_I, [_P, _P, _L, _I, _I, _P, _I]),
    "cs_scaler_pca": (_I, [_P, _P, _L, _I, _P, _I]),
    "cs_svm_decision": (_I, [_P, _P, _L, _I, _P, _P, _I]),
    "cs_preproc_create": (_I, [_I, C.POINTER(_P)]),
    "cs_preproc_free": (None, [_P]),
    "cs_preproc_wait_stream": (_I, [_P, _P]),
    "cs_preprocess": (_I, [_P, _P, _I, _L, _I, _P, _P, _P, _L, C.c_double, _P, _P, _I]),
    "cs_preproc_last_timing": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_L)]),
    "cs_fit_create": (_I, [_I, C.POINTER(_P)]),
    "cs_fit_free": (None, [_P]),
    "cs_fit_wait_stream": (_I, [_P, _P]),
    "cs_fit_scaler": (_I, [_P, _P, _L, C.c_int32, _I, _P, _P]),
    "cs_fit_pca_moments": (_I, [_P, _P, _L, C.c_int32, _I, _P, _P, _P, _P]),
    "cs_fit_project": (_I, [_P, _P, _L, C.c_int32, _I, _P, _P, _P, _P, C.c_int32, _P]),
    "cs_fit_ocsvm": (_I, [_P, _P, _L, C.c_int32, C.c_double, C.c_double, C.c_double, _L, _P, C.POINTER(C.c_double),
                          C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_int32)]),
    "cs_fit_last_ms": (_I, [_P, C.POINTER(C.c_double)]),
    "cs_synth_crops": (_I, [_P, C.c_uint64, _L, _L, C.c_int32, _P]),
    "cs_profile_enable": (_I, [_P, _I]),
    "cs_profile_reset": (_I, [_P]),
    "cs_profile_kernel_count": (_I, []),
    "cs_profile_kernel_name": (C.c_char_p, [_I]),
    "cs_profile_get": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(_L), C.POINTER(C.c_double)]),
    "cs_profile_mfma_per_cell": (_I, [_P, _I, C.POINTER(C.c_double)]),
    "cs_profile_bf16_mfma_per_cell": (_I, [_P, _I, C.POINTER(C.c_double)]),
    "cs_train_param_count": (_I, [C.POINTER(_L), C.POINTER(_L)]),
    "cs_train_param_count_of": (_I, [_P, C.POINTER(_L), C.POINTER(_L)]),
    "cs_train_create": (_I, [C.POINTER(CSCaeWeights), C.POINTER(CSTrainCfg), _I, C.POINTER(_P)]),
    "cs_train_free": (None, [_P]),
    "cs_train_wait_stream": (_I, [_P, _P]),
    "cs_train_step": (_I, [_P, _P, _P, _L, _I, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cs_train_step_async": (_I, [_P, _P, _P, _L, _I, C.c_float]),
    "cs_train_inputs_consumed": (_I, [_P, _P]),
    "cs_train_draw_transforms": (_I, [C.POINTER(CSAugConfig), C.c_uint64, C.c_uint64, _L, C.c_int32, C.c_int32, _P]),
    "cs_train_fit_step": (_I, [_P, _P, _L, _P, _L, C.POINTER(CSAugConfig), C.c_uint64, C.c_uint64, C.c_float]),
    "cs_train_read_metrics": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L), _I]),
    "cs_train_forward_backward": (_I, [_P, _P, _P, _L, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cs_train_apply": (_I, [_P, C.c_float]),
    "cs_train_set_grad_buffer": (_I, [_P, _P]),
    "cs_train_set_sync_bn": (_I, [_P, _P, _P, _P, _L, _I, _I]),
    "cs_train_eval": (_I, [_P, _P, _P, _L, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cs_train_augment": (_I, [_P, _P, _L, _P, _P, _I]),
    "cs_train_export": (_I, [_P, _P, _P, _P]),
    "cs_train_tensor": (_I, [_P, _I, _I, _L, _P]),
    "cs_train_import": (_I, [_P, _P, _P]),
}

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libcellscreen.so and bind every symbol of the header.  torch (if importable) is
    imported first so the process holds ONE HIP runtime: torch bundles libamdhip64.so.7 and
    the dynamic linker then resolves our NEEDED entry to that already-loaded copy."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built; run `python cell-image-analysis_amd/build.py` "
                                "(or __graft_entry__.build()).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (plumbing only: shares its HIP runtime)
    except Exception:
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.cs_abi_version() != 2:
        raise RuntimeError("libcellscreen ABI version mismatch")
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        lib = load_library()
        msg = lib.cs_last_error().decode(errors="replace") or lib.cs_status_string(status).decode()
        raise CellScreenError(status, msg)


def _ptr(a) -> Optional[int]:
    """Address of a numpy array (host) or anything exposing data_ptr() (torch tensor)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(f"cannot take the address of {type(a)}")


def order_after_torch(wait_fn, handle, *buffers):
    """Device buffers handed to the library may still be being produced on torch's current stream (a gather, a
    cast, an RCCL collective torch has already ordered its stream after); the handle works on its own non-blocking
    stream.  Make that stream wait for torch's -- an event dependency on the device, no host synchronisation.  A
    no-op when no buffer is a CUDA tensor (include/cellscreen.h, "Device INPUTS")."""
    dev = None
    for b in buffers:
        if b is not None and not isinstance(b, np.ndarray) and getattr(b, "is_cuda", False):
            dev = b.device
            break
    if dev is None:
        return
    import torch
    check(wait_fn(handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def order_torch_after(signal_fn, handle, *buffers):
    """The other direction: torch's current stream waits for the point the library has recorded on its own stream (the input
    copies of an asynchronous training step, the kernel behind a device-side augmentation), so that torch neither reads a
    result early nor hands the memory of a batch the library has not consumed yet to a later allocation."""
    dev = None
    for b in buffers:
        if b is not None and not isinstance(b, np.ndarray) and getattr(b, "is_cuda", False):
            dev = b.device
            break
    if dev is None:
        return
    import torch
    check(signal_fn(handle, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def mem_kind(a) -> int:
    if a is None or isinstance(a, np.ndarray):
        return CS_MEM_HOST
    if hasattr(a, "is_cuda"):
        return CS_MEM_DEVICE if a.is_cuda else CS_MEM_HOST
    raise TypeError(f"unsupported buffer type {type(a)}")
