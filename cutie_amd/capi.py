"""ctypes binding of libcutie_hip.so (the C ABI declared in include/cutie_hip.h).

PyTorch-ROCm supplies device memory and streams only; every entry point takes raw device
pointers and the current HIP stream.  There is no CPU fallback: if the library is missing or
an entry point fails, a CutieError is raised.
"""
import ctypes
import os
import threading

import torch

F32, BF16, F16 = 0, 1, 2
ACT_FROM_Y = 16      # CUTIE_ACT_FROM_Y
MAX_SEGS = 6
REDUCE_SCRATCH = 1024
OPT_SCRATCH = 2048
LOSS_SCALE_STATE = 8

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUTIE_HIP_LIB") or os.path.join(_HERE, "libcutie_hip.so")   # override: A/B builds


class CutieError(RuntimeError):
    pass


class ConvSeg(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("w_tap_stride", ctypes.c_int64),
                ("ld_x", ctypes.c_int), ("cin", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int),
                ("dil", ctypes.c_int), ("flip", ctypes.c_int)]


class PackDesc(ctypes.Structure):
    _fields_ = [("w_hwio", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_dgrad", ctypes.c_void_p),
                ("taps", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
                ("cin_pad", ctypes.c_int), ("cout_pad", ctypes.c_int)]


class WgradDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("ld_x", ctypes.c_int), ("cin", ctypes.c_int),
                ("dz", ctypes.c_void_p), ("ld_dz", ctypes.c_int), ("cout", ctypes.c_int),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int),
                ("dil", ctypes.c_int), ("dw", ctypes.c_void_p), ("accumulate", ctypes.c_int)]


class SumRowsDesc(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("C", ctypes.c_int)]


MAX_BATCH = 32
POOL_TILES = 32

_c = ctypes
_vp, _i, _i64, _f, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); must list every symbol of include/cutie_hip.h
SIGNATURES = {
    "cutie_version": (_i, []),
    "cutie_last_error": (_c.c_char_p, []),
    "cutie_grid_build_workspace_bytes": (_sz, [_i]),
    "cutie_grid_build": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _sz, _vp]),
    "cutie_embed_fwd": (_i, [_vp, _i64, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "cutie_embed_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "cutie_embed_bwd": (_i, [_vp, _i64, _vp, _i, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "cutie_conv_stats_tiles": (_i, [_i, _i]),
    "cutie_conv_fwd": (_i, [_c.POINTER(ConvSeg), _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "cutie_conv_in_fwd": (_i, [_c.POINTER(ConvSeg), _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _sz,
                               _vp, _c.POINTER(_i), _i, _vp]),
    "cutie_conv_dgrad_in_stats": (_i, [_c.POINTER(ConvSeg), _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _f, _vp,
                                       _i, _vp, _i, _i, _vp]),
    "cutie_conv_sync_bytes": (_sz, [_i, _i, _i]),
    "cutie_conv_clock_probe": (_i, [_vp]),
    "cutie_conv_dgrad_in_bwd": (_i, [_c.POINTER(ConvSeg), _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i,
                                     _vp, _vp, _i, _vp, _i, _vp, _sz, _vp, _i, _vp]),
    "cutie_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "cutie_conv_wgrad": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "cutie_conv_wgrad_batched_workspace_bytes": (_sz, [_c.POINTER(WgradDesc), _i, _i]),
    "cutie_conv_wgrad_batched": (_i, [_c.POINTER(WgradDesc), _i, _i, _vp, _sz, _vp]),
    "cutie_pack_weights": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "cutie_pack_weights_batched": (_i, [_c.POINTER(PackDesc), _i, _i, _vp]),
    "cutie_in_tiles": (_i, [_i, _i, _i]),
    "cutie_in_stats": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "cutie_in_apply": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "cutie_in_bwd_stats": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "cutie_in_bwd_apply": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _f, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i,
                                _vp, _vp]),
    "cutie_sum_rows": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "cutie_sum_rows_batched": (_i, [_c.POINTER(SumRowsDesc), _i, _i, _i, _vp]),
    "cutie_spatial_mean_bcast": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cutie_softmax_ce_fwd": (_i, [_vp, _i, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "cutie_hnm_workspace_bytes": (_sz, [_i64]),
    "cutie_hnm_weights": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    "cutie_hnm_weights_static": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "cutie_softmax_ce_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "cutie_head_tiles": (_i, [_i]),
    "cutie_head_chunks": (_i, [_i]),
    "cutie_head_workspace_bytes": (_sz, [_i, _i]),
    "cutie_head_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp,
                            _vp, _sz, _vp]),
    "cutie_head_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp,
                            _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "cutie_ghm_state_bytes": (_sz, [_i]),
    "cutie_ghm_weights": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _f, _f, _vp, _sz, _vp, _vp, _vp]),
    "cutie_l2_add": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _vp]),
    "cutie_sumsq": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "cutie_loss_scale_init": (_i, [_vp, _f, _vp]),
    "cutie_loss_scale_update": (_i, [_vp, _vp, _f, _f, _i, _f, _f, _vp, _vp]),
    "cutie_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp, _f, _f, _i, _vp, _vp, _vp]),
    "cutie_opt_norm": (_i, [_vp, _vp, _i64, _i64, _f, _vp, _i64, _i, _vp, _vp, _vp]),
    "cutie_opt_step": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _f, _f, _c.POINTER(PackDesc), _i, _i, _f, _f, _f, _f, _f, _f, _i,
                            _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "cutie_batch_gather": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "cutie_batch_scatter_add": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "cutie_convert": (_i, [_vp, _i, _i, _vp, _i, _i, _i64, _i, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises CutieError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CutieError("%s is missing: build it with `python -m cutie_amd.build` (hipcc, gfx950). "
                             "There is no CPU fallback." % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise CutieError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(rc, name):
    if rc != 0:
        msg = lib().cutie_last_error()
        raise CutieError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))


_tls = threading.local()      # .dev: device of the tensor most recently handed to ptr() on this thread


def stream(device=None):
    """The HIP stream a call's kernels go to: torch's current stream OF THE DEVICE THAT OWNS THE CALL'S TENSORS (the
    device of the last tensor given to ptr() on this thread - every binding evaluates its ptr() arguments before its
    trailing stream()), not of whatever device happens to be current on the calling thread.  `device` overrides, for
    calls whose pointers sit inside a descriptor array."""
    if device is not None:
        _tls.dev = torch.device(device)
    dev = getattr(_tls, "dev", None)
    return torch.cuda.current_stream(dev).cuda_stream


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise CutieError("unsupported activation dtype %s" % t.dtype)


def torch_dtype(dt):
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[dt]


def parse_dtype(dtype):
    """'f32' / 'bf16' / 'f16' (or a torch dtype, or the C constant) -> C constant"""
    table = {"f32": F32, "fp32": F32, "float32": F32, torch.float32: F32,
             "bf16": BF16, "bfloat16": BF16, torch.bfloat16: BF16,
             "f16": F16, "fp16": F16, "float16": F16, "half": F16, torch.float16: F16}
    if isinstance(dtype, int) and not isinstance(dtype, bool) and dtype in (F32, BF16, F16):
        return dtype
    if dtype not in table:
        raise CutieError("unknown dtype %r (f32, bf16, f16)" % (dtype,))
    return table[dtype]


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise CutieError("expected a device tensor")
    _tls.dev = t.device
    return t.data_ptr()


def ld_of(t):
    """leading dimension (elements) of a [..., C] view whose rows are dense in the last axis
    and whose leading axes collapse to one row index"""
    if t.stride(-1) != 1:
        raise CutieError("channel axis must be contiguous")
    ld = t.stride(-2)
    for d in range(t.dim() - 2):
        if t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
            raise CutieError("view is not row-collapsible: shape %s strides %s" % (tuple(t.shape), t.stride()))
    return ld


def call(name, *args):
    """Runs an entry point with the device of its tensors current (per-device library state - dynamic-LDS grants,
    the static work area of cutie_hnm_weights_static - belongs to the CURRENT device)."""
    fn = getattr(lib(), name)
    dev = getattr(_tls, "dev", None)
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = fn(*args)
    else:
        rc = fn(*args)
    _check(rc, name)
