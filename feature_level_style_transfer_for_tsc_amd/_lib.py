"""ctypes binding of libfst_hip.so (the C ABI declared in include/fst_hip.h).

There is no CPU fallback: every op in this package goes through these symbols, and a missing or
stale library raises immediately (``FstLibraryError``) instead of silently computing elsewhere.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch

ABI_VERSION = 12
_LIB_NAME = "libfst_hip.so"
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FST_HIP_LIB", os.path.join(_HERE, _LIB_NAME))     # override: diagnostic builds only


class FstLibraryError(RuntimeError):
    pass


class WSrc(Structure):
    """mirror of ``fst_wsrc``"""
    _fields_ = [("w", c_void_p), ("off0", c_int64), ("sm", c_int64), ("sc", c_int64), ("st", c_int64)]


_P = c_void_p          # device pointer
_I32P = c_void_p       # int32 device/host pointers are passed as raw addresses too
_SIGNATURES = {
    "fst_version": (c_int, []),
    "fst_last_error": (c_char_p, []),
    "fst_pack_weights": (c_int, [_I32P, _I32P, c_int, POINTER(WSrc), POINTER(WSrc), c_int, c_int, c_int, c_int, _P, c_void_p]),
    "fst_pack_weights_bf16x3": (c_int, [_I32P, _I32P, c_int, POINTER(WSrc), POINTER(WSrc), c_int, c_int, c_int, c_int, _P, c_void_p]),
    "fst_unpack_weights": (c_int, [_I32P, _I32P, c_int, _P, c_int, _P, c_int64, c_int64, c_int64, c_int64,
                                   _P, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "fst_mask_taps": (c_int, [_P, _I32P, _I32P, c_int, c_int, c_int, c_void_p]),
    "fst_conv_gemm": (c_int, [_P, c_int64, _P, c_int64, _P, _I32P, _I32P, c_int, _P, _P, c_int64, _P, c_int64,
                              _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fst_conv_wgrad": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int, _P, _I32P, _I32P, c_int,
                               c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_row_sum": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_void_p]),
    "fst_bn_stats": (c_int, [_P, c_int, c_int, c_int, _P, c_int64, c_void_p]),
    "fst_bn_finalize": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_void_p]),
    "fst_bn_apply": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_bn_bwd_reduce": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, c_void_p]),
    "fst_bn_bwd_apply": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_gate_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_gate_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_coupling_sum_slots": (c_int64, [c_int, c_int, c_int]),
    "fst_coupling_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int64, _P, c_void_p]),
    "fst_coupling_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_coupling_inv_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_coupling_inv_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_rmsprop_multi": (c_int, [_P, _P, _P, _P, _P, c_int, c_float, c_float, c_void_p]),
    "fst_adam_multi": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_float, c_float, c_float, c_float, c_void_p]),
    "fst_wn_stack_fwd_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "fst_wn_stack_fwd": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_wn_stack_bwd_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "fst_wn_stack_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_wn_wgrad_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "fst_wn_wgrad_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "fst_wn_wgrad_in": (c_int, [_P, _P, _P, c_int, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                                c_void_p]),
    "fst_wn_wgrad_rs": (c_int, [_P, _P, _P, c_int, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_nt_gemm_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "fst_nt_gemm": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, c_void_p]),
    "fst_tap_wgrad_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "fst_tap_wgrad_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "fst_tap_wgrad": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "fst_dense_tap_wgrad_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "fst_dense_tap_wgrad_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "fst_dense_tap_wgrad": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "fst_relu_bwd": (c_int, [_P, _P, _P, c_int64, c_void_p]),
    "fst_gemm_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "fst_gemm": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, c_int64, c_int, c_int, c_int, _P, c_int, c_float, _P, c_int64,
                         c_void_p]),
    "fst_act_bwd": (c_int, [_P, _P, _P, c_int64, c_float, c_void_p]),
    "fst_batch_sum": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_void_p]),
    "fst_noise_transfer_fwd": (c_int, [_P, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_void_p]),
    "fst_bcast_add": (c_int, [_P, _P, _P, c_int, c_int64, c_void_p]),
    "fst_noise_transfer_bwd": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_void_p]),
    "fst_noise_transfer_dw": (c_int, [_P, _P, _P, _P, c_int, c_int, c_void_p]),
    "fst_noise_transfer_bwd_apply": (c_int, [_P, _P, _P, _P, c_float, c_float, c_int, _P, _P, c_int64, c_void_p]),
    "fst_wn_fold_fwd": (c_int, [_P, c_int, _P, _P, c_void_p]),
    "fst_wn_fold_bwd": (c_int, [_P, c_int, _P, _P, _P, c_void_p]),
    "fst_logdet_inv": (c_int, [_P, c_int, _P, _P, c_void_p]),
    "fst_wn_image_bytes": (c_int64, [c_int, c_int]),
    "fst_wn_pack": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, c_void_p]),
    "fst_wn_layer_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int64, c_void_p]),
    "fst_wn_bwd_image_bytes": (c_int64, [c_int, c_int]),
    "fst_wn_pack_bwd": (c_int, [_P, c_int, c_int, c_int, _P, c_int64, c_void_p]),
    "fst_wn_layer_bwd": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_wn_dgrad_image_bytes": (c_int64, [c_int]),
    "fst_wn_pack_dgrad": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int64, c_void_p]),
    "fst_wn_dgrad_fits": (c_int, [c_int, c_int, c_int]),
    "fst_wn_layer_dgrad": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                   c_void_p]),
    "fst_gru_fwd": (c_int, [_P, _P, _P, _P, _P, _I32P, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_gru_bwd": (c_int, [_P, _P, _P, _P, _I32P, c_int, _P, _P, c_int, c_int, c_int, c_int64, c_void_p]),
    "fst_lstm2_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int64, c_void_p]),
    "fst_lstm2_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int64, c_void_p]),
    "fst_axpy": (c_int, [_P, _P, c_float, c_int64, c_void_p]),
    "fst_add_slices": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, c_int, c_void_p]),
    "fst_cpc_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "fst_cpc_nce_slots": (c_int64, [c_int, c_int, c_int, c_int]),
    "fst_cpc_nce_fwd": (c_int, [_P, c_int64, c_int64, c_int64, _I32P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_void_p]),
    "fst_cpc_nce_bwd": (c_int, [_P, c_int64, c_int64, c_int64, _I32P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                                c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[ctypes.CDLL] = None


def load(path: Optional[str] = None) -> ctypes.CDLL:
    """Load (once) and return the shared library; raise FstLibraryError if it is missing or stale."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FstLibraryError(
            f"{p} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:                                          # e.g. libamdhip64 missing
        raise FstLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise FstLibraryError(f"{p} does not export {name}; rebuild the library") from e
        fn.restype, fn.argtypes = res, args
    if lib.fst_version() != ABI_VERSION:
        raise FstLibraryError(f"{p}: ABI version {lib.fst_version()} != expected {ABI_VERSION}; rebuild")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fst_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def stream_ptr() -> int:
    """The HIP stream torch is currently enqueuing on — kernels are launched on the same stream."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise FstLibraryError(f"{name} lives on {t.device}; the fst ops run on an MI355X only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
