"""ctypes binding of libpnpp_hip.so (the C ABI declared in include/pnpp_hip.h).

There is deliberately no fallback: if the shared library is missing or cannot be loaded, every
entry point raises.  PyTorch is used only for device memory, streams and autograd bookkeeping.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpnpp_hip.so")

PNPP_MAX_LAYERS = 4
PNPP_OK, PNPP_ERR_ARG, PNPP_ERR_RANGE, PNPP_ERR_LAUNCH, PNPP_ERR_WORKSPACE = 0, -1, -2, -3, -4
NORM_NONE, NORM_BATCH, NORM_LAYER = 0, 1, 2

_fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class SaDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("S", C.c_int), ("K", C.c_int), ("D", C.c_int), ("L", C.c_int),
                ("C", C.c_int * PNPP_MAX_LAYERS), ("group_all", C.c_int), ("training", C.c_int),
                ("eps", C.c_float), ("momentum", C.c_float)]


_PL = _fp * PNPP_MAX_LAYERS


class SaFwdArgs(C.Structure):
    _fields_ = [("xyz", _fp), ("points", _fp), ("centre_idx", _fp), ("neighbour_idx", _fp),
                ("conv_w", _PL), ("conv_b", _PL), ("bn_w", _PL), ("bn_b", _PL), ("bn_rm", _PL), ("bn_rv", _PL), ("bn_nbt", _PL),
                ("new_xyz", _fp), ("out", _fp), ("saved", _fp), ("scratch", _fp)]


class SaBwdArgs(C.Structure):
    _fields_ = [("xyz", _fp), ("points", _fp), ("conv_w", _PL), ("bn_w", _PL), ("bn_b", _PL),
                ("dout", _fp), ("saved", _fp), ("scratch", _fp),
                ("d_conv_w", _PL), ("d_conv_b", _PL), ("d_bn_w", _PL), ("d_bn_b", _PL), ("dpoints", _fp)]


class FcDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("K", C.c_int), ("N", C.c_int), ("norm", C.c_int), ("relu", C.c_int),
                ("training", C.c_int), ("eps", C.c_float), ("momentum", C.c_float), ("drop_scale", C.c_float)]


class FcFwdArgs(C.Structure):
    _fields_ = [("x", _fp), ("w", _fp), ("b", _fp), ("nw", _fp), ("nb", _fp), ("rm", _fp), ("rv", _fp), ("nbt", _fp),
                ("mask", _fp), ("mask_out", _fp), ("drop_p", C.c_float), ("rng_seed", C.c_uint64), ("rng_counter", _fp),
                ("y", _fp), ("saved", _fp), ("scratch", _fp)]


class FcBwdArgs(C.Structure):
    _fields_ = [("x", _fp), ("w", _fp), ("b", _fp), ("nw", _fp), ("nb", _fp), ("mask", _fp), ("dy", _fp),
                ("saved", _fp), ("scratch", _fp), ("dx", _fp), ("dw", _fp), ("db", _fp), ("dnw", _fp), ("dnb", _fp)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
_i, _f, _u64, _sz = C.c_int, C.c_float, C.c_uint64, C.c_size_t
# int fn(double *buf, size_t n_doubles, void *stream, void *user): sums buf over the ranks in place (SyncBN, pnpp_set_stats_exchange)
STATS_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
SIGNATURES = {
    "pnpp_last_error": (C.c_char_p, []),
    "pnpp_abi_version": (_i, []),
    "pnpp_profile_enable": (_i, [_i]),
    "pnpp_profile_report": (_i, [C.c_char_p, _sz]),
    "pnpp_square_distance": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp]),
    "pnpp_knn": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "pnpp_fps": (_i, [_fp, _i, _i, _i, _fp, _fp, _fp]),
    "pnpp_ball_query": (_i, [_fp, _fp, _i, _i, _i, _f, _i, _fp, _fp]),
    "pnpp_sample_random": (_i, [_u64, _u64, _i, _i, _i, _fp, _fp]),
    "pnpp_subsample_points": (_i, [_u64, _u64, _fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "pnpp_sample_random_dev": (_i, [_u64, _fp, _u64, _i, _i, _i, _fp, _fp]),
    "pnpp_sample_random_dev2": (_i, [_u64, _fp, _u64, _i, _i, _i, _fp, _i, _i, _fp, _fp]),
    "pnpp_index_points": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "pnpp_index_points_bwd": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "pnpp_sa_saved_bytes": (_sz, [C.POINTER(SaDesc)]),
    "pnpp_sa_scratch_bytes": (_sz, [C.POINTER(SaDesc)]),
    "pnpp_sa_forward": (_i, [C.POINTER(SaDesc), C.POINTER(SaFwdArgs), _fp]),
    "pnpp_sa_backward": (_i, [C.POINTER(SaDesc), C.POINTER(SaBwdArgs), _fp]),
    "pnpp_sa_saved_neighbours": (_fp, [C.POINTER(SaDesc), _fp]),
    "pnpp_sa_group_pair": (_i, [C.POINTER(SaDesc), C.POINTER(SaDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "pnpp_sa_saved_argmax": (_fp, [C.POINTER(SaDesc), _fp]),
    "pnpp_sa_saved_relu_mask": (_i, [C.POINTER(SaDesc), _fp, _fp, _fp, _i, _fp, _fp]),
    "pnpp_build_flags": (C.c_uint, []),
    "pnpp_debug_wsd3_timeouts": (_i, []),
    "pnpp_fc_saved_bytes": (_sz, [C.POINTER(FcDesc)]),
    "pnpp_fc_scratch_bytes": (_sz, [C.POINTER(FcDesc)]),
    "pnpp_fc_forward": (_i, [C.POINTER(FcDesc), C.POINTER(FcFwdArgs), _fp]),
    "pnpp_fc_backward": (_i, [C.POINTER(FcDesc), C.POINTER(FcBwdArgs), _fp]),
    "pnpp_vm_head_kl": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp]),
    "pnpp_vm_fc_head_kl_step": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "pnpp_vm_fc_head_kl_step_sample": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _u64, _fp, _u64, _i, _i, _i, _fp, _i, _i,
                                            _fp, _fp]),
    "pnpp_vm_head_kl_mean": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    "pnpp_vm_head_bwd": (_i, [_fp, _fp, _fp, _i, _fp, _fp]),
    "pnpp_vm_kl_single": (_i, [_fp, _fp, _fp, _fp, _i, _fp, _fp, _fp, _fp]),
    "pnpp_vm_match_loss": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp]),
    "pnpp_mvm_fc_head_match_step": (_i, [_fp] * 9 + [_i, _i, _i, _f, _f] + [_fp] * 11 + [_u64, _fp, _u64, _i, _i, _i, _fp, _i, _i, _fp, _fp]),
    "pnpp_mvm_head": (_i, [_fp, _fp, _fp, _i, _i, _f, _f, _fp, _fp, _fp, _fp]),
    "pnpp_mvm_head_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _f, _f, _fp, _fp, _fp, _fp]),
    "pnpp_soft_ce": (_i, [_fp, _fp, _i, _i, _fp, _fp, _fp]),
    "pnpp_l2_normalize": (_i, [_fp, _i, _i, _f, _fp, _fp]),
    "pnpp_l2_normalize_bwd": (_i, [_fp, _fp, _i, _i, _f, _fp, _fp]),
    "pnpp_mse": (_i, [_fp, _fp, _sz, _fp, _fp, _fp]),
    "pnpp_mse_rows": (_i, [_fp, _fp, _i, _i, _fp, _fp, _fp]),
    "pnpp_orth_loss": (_i, [_fp, _fp, _i, _i, _fp, _fp, _fp, _fp]),
    "pnpp_proj_probs": (_i, [_fp, _fp, _i, _i, _fp, _fp]),
    "pnpp_proj_probs_bwd": (_i, [_fp, _fp, _fp, _i, _i, _fp, _fp]),
    "pnpp_linear_smallk": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "pnpp_attention_fwd": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _f, _fp, _fp, _fp]),
    "pnpp_attention_dropout_mask": (_i, [_u64, _u64, _i, _i, _i, _f, _fp, _fp, _fp]),
    "pnpp_attention_dropout_mask_dev": (_i, [_u64, _fp, _u64, _i, _i, _i, _f, _fp, _fp, _fp]),
    "pnpp_add_layernorm": (_i, [_fp, _fp, _fp, _fp, _i, _i, _f, _fp, _fp]),
    "pnpp_mean_points": (_i, [_fp, _i, _i, _i, _fp, _fp]),
    "pnpp_linear_smallk_bwd_scratch_bytes": (_sz, [_i, _i]),
    "pnpp_linear_smallk_bwd": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp]),
    "pnpp_attention_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _f, _fp, _fp, _fp]),
    "pnpp_add_layernorm_bwd_scratch_bytes": (_sz, [_i, _i]),
    "pnpp_add_layernorm_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _f, _fp, _fp, _fp, _fp]),
    "pnpp_mean_points_bwd": (_i, [_fp, _i, _i, _i, _fp, _fp]),
    "pnpp_set_matmul_precision": (_i, [_i]),
    "pnpp_set_stats_exchange": (_i, [STATS_EXCHANGE_FN, C.c_void_p, C.c_void_p, _sz]),
    "pnpp_stats_exchange_enabled": (_i, []),
    "pnpp_get_matmul_precision": (_i, []),
    "pnpp_set_split_products": (_i, [_i]),
    "pnpp_get_split_products": (_i, []),
    "pnpp_adam_step": (_i, [_fp, _fp, _fp, _fp, _sz, _i, _f, _f, _f, _f, _f, _fp]),
    "pnpp_adam_step_zero": (_i, [_fp, _fp, _fp, _fp, _sz, _i, _f, _f, _f, _f, _f, _fp]),
    "pnpp_adam_step_dev": (_i, [_fp, _fp, _fp, _fp, _sz, _fp, _f, _f, _f, _f, _f, _i, _fp]),
    "pnpp_adam_step_clip": (_i, [_fp, _fp, _fp, _fp, _sz, _i, _f, _f, _f, _f, _f, _fp, _f, _i, _fp]),
    "pnpp_adam_step_dev_clip": (_i, [_fp, _fp, _fp, _fp, _sz, _fp, _f, _f, _f, _f, _f, _fp, _f, _i, _fp]),
    "pnpp_sumsq": (_i, [_fp, _sz, _fp, _fp, _sz, _fp]),
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """The loaded shared library (loads on first use; never falls back to anything else)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} is missing: build it with `python -m pnpp_hip.build` (hipcc, gfx950). "
                "This package has no CPU or PyTorch fallback.")
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise HipExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.pnpp_abi_version() != 5:
            raise HipExtensionMissing("libpnpp_hip.so ABI version mismatch; rebuild it")
        _lib = h
    return _lib


def last_error() -> str:
    return lib().pnpp_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Turns a pnpp_status into the exception type the reference's Python surface would raise."""
    if rc == PNPP_OK:
        return
    msg = last_error()
    if rc == PNPP_ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError(msg)
