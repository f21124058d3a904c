"""ctypes binding of libbmp_hip.so (the C ABI declared in include/bmp.h).

The HIP library is the product path: there is no CPU or PyTorch fallback.  A missing or
unloadable library raises immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMP_LIB_PATH") or os.path.join(_HERE, "libbmp_hip.so")      # BMP_LIB_PATH: A/B builds (tools)

_P, _I, _Z, _F = c_void_p, c_int, c_size_t, c_float

# name -> (restype, argtypes); mirrors include/bmp.h declaration by declaration
SIGNATURES = {
    "bmp_version": (_I, []),
    "bmp_tile_rows": (_I, []),
    "bmp_prof_start": (_I, [_I]),
    "bmp_prof_stop": (_I, [_P]),
    "bmp_prof_collect": (_I, [_P, _P, _P, _P, _P, _I]),
    "bmp_stream_create_low": (_I, [_P]),
    "bmp_stream_destroy": (_I, [_P]),
    "bmp_embed_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "bmp_embed_bwd_ws_floats": (_Z, [_I, _I, _I]),
    "bmp_embed_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _Z, _P]),
    "bmp_msg_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    "bmp_msg_bwd_ws_floats": (_Z, [_I, _I, _I]),
    "bmp_msg_bwd": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _Z, _P, _P]),
    "bmp_gru_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "bmp_gru_bwd_ws_floats": (_Z, [_I, _I]),
    "bmp_gru_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _Z, _P, _P]),
    "bmp_gru_state_fwd": (_I, [_P, _P, _P, _I, _I] + [_P] * 8),
    "bmp_gru_state_bwd_ws_floats": (_Z, [_I, _I]),
    "bmp_gru_state_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I] + [_P] * 11 + [_Z, _P]),
    "bmp_ggnn_step_supported": (_I, [_I]),
    "bmp_ggnn_step_fwd": (_I, [_P, _I, _I, _I, _I] + [_P] * 14 + [_I, _I, _P]),
    "bmp_ggnn_step_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I] + [_P] * 10 + [_I, _I, _I, _P]),
    "bmp_step_wgrad_lists_used": (_I, [_I, _I]),
    "bmp_ggnn_step_wgrad_ws_floats": (_Z, [_I, _I]),
    "bmp_ggnn_step_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "bmp_type_rows_ws_ints": (_Z, [_I]),
    "bmp_type_rows": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "bmp_type_rows_live": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "bmp_readout_tile_supported": (_I, [_I, _I, _I]),
    "bmp_readout_tile_fwd": (_I, [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "bmp_relgcn_layer_supported": (_I, [_I, _I]),
    "bmp_relgcn_layer_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "bmp_relgcn_layer_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "bmp_relgcn_layer_wgrad_ws_floats": (_Z, [_I, _I]),
    "bmp_relgcn_layer_wgrad": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "bmp_readout_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P]),
    "bmp_readout_bwd_ws_floats": (_Z, [_I, _I, _I, _I]),
    "bmp_readout_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _P, _P]),
    "bmp_linear_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P]),
    "bmp_wgrad_ws_floats_c": (_Z, [_I, _I, _I]),
    "bmp_linear_wgrad": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "bmp_segpool_fwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _P]),
    "bmp_segpool_bwd": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "bmp_segsoftmax_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "bmp_segsoftmax_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "bmp_rowbcast_fwd": (_I, [_P, _I, _P, _I, _P, _P]),
    "bmp_rowbcast_bwd": (_I, [_P, _I, _P, _P, _I, _P, _P]),
    "bmp_rowdot_fwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P]),
    "bmp_rowdot_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "bmp_rowcorr_fwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _P]),
    "bmp_rowcorr_bwd": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P]),
    "bmp_dense_count": (_I, [_P, _I, _I, _P, _P, _P]),
    "bmp_dense_to_csr": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P]),
    "bmp_collate_plan": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P]),
    "bmp_collate_pair_meta": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "bmp_collate_emit": (_I, [_P, _I] + [_P] * 16 + [_P, _P]),
    "bmp_collate_plan_enc": (_I, [_P, _P, _I, _P, _I, _I, _I, _I] + [_P] * 11),
    "bmp_bimpm_supported": (_I, [_I, _I, _I]),
    "bmp_bimpm_ws_floats": (_Z, [_I, _I, _I, _I, _I]),
    "bmp_bimpm_fwd": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bmp_bimpm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bmp_encrows_expand": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "bmp_encrows_reduce": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "bmp_rescale_adj": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "bmp_mlp_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "bmp_mlp_bwd_ws_floats": (_Z, [_I, _I, _P]),
    "bmp_mlp_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "bmp_mlp_sce_ws_floats": (_Z, [_I]),
    "bmp_mlp_sce_fwdbwd": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "bmp_mlp_bwd_w": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bmp_sce_fwd": (_I, [_P, _P, _I, _P, _P, _P]),
    "bmp_sce_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "bmp_pairfeat_cols": (_I, [_I, _I, _I]),
    "bmp_pairfeat_fwd": (_I, [_I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "bmp_pairfeat_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "bmp_gather_sum": (_I, [_P, _I, _P, _P, _I, _I, _P]),
    "bmp_adam_step": (_I, [_P, _P, _P, _P, _I, _F, _P, _F, _F, _F, _F, _F, _P]),
    "bmp_coattn_zcols": (_I, [_I, _I]),
    "bmp_coattn_nie_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I] + [_P] * 7 + [_I, _P, _I, _I, _I, _I, _I, _I] + [_P] * 17 + [_P, _Z, _P]),
    "bmp_coattn_big_ws_floats": (_Z, [_I] * 5),
    "bmp_coattn_nie_bwd_ws_floats": (_Z, [_I] * 8),
    "bmp_coattn_nie_bwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I] + [_P] * 7 + [_I, _P, _I, _I, _I, _I, _I, _I] + [_P] * 21 + [_Z, _P, _P, _P, _P, _P]),
}

_lib = None
_hip = None


class BmpLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; fail loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BmpLibraryError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no fallback path.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:                                  # pragma: no cover
            raise BmpLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise BmpLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _hip_error_string(code: int) -> str:
    global _hip
    try:
        if _hip is None:
            _hip = ctypes.CDLL("libamdhip64.so")
            _hip.hipGetErrorString.restype = ctypes.c_char_p
            _hip.hipGetErrorString.argtypes = [c_int]
        return _hip.hipGetErrorString(code).decode()
    except Exception:                                         # pragma: no cover
        return "?"


def check(rc: int, name: str) -> None:
    if rc == 0:
        return
    if rc < -1000:
        raise ValueError(f"{name}: argument check failed at csrc line {-(rc + 1000)}")
    raise RuntimeError(f"{name}: HIP error {rc} ({_hip_error_string(rc)})")


def ptr(t: torch.Tensor | None):
    """Device address for a ``c_void_p`` argument (a plain int: ctypes converts it, and a step makes ~180 of these calls)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Handle of torch's current stream on the current device (every launch asks: the raw accessor costs a fraction of a
    microsecond, torch.cuda.current_stream() ten)."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_rows(t: torch.Tensor, name: str, cols: int | None = None) -> torch.Tensor:
    """Shape/dtype/contiguity checks before a launch (ValueError, like the reference's
    option checks, e.g. models/ggnn.py:250)."""
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError(f"{name}: expected a float32 CUDA(HIP) tensor")
    if t.dim() != 2 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous 2-d tensor, got shape {tuple(t.shape)}")
    if cols is not None and t.shape[1] != cols:
        raise ValueError(f"{name}: expected {cols} columns, got {t.shape[1]}")
    return t
