"""ctypes binding of libuig.so (the C ABI declared in include/uig.h).  Fails loudly when the library is missing:
there is no CPU / PyTorch fallback for any op on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UIG_LIB_PATH") or os.path.join(_HERE, "libuig.so")   # UIG_LIB_PATH: A/B a second build of the same ABI
_lib = None

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
GATHER_DIRECT, GATHER_TRANSPOSED = 0, 1
PACK_ROW_DIM0, PACK_ROW_DIM1 = 0, 1
K_NONE, K_IGEMM, K_STRIP128, K_STRIP256, K_STRIP_PK, K_ROWSTRIP, K_HEADROW, K_GEMV, K_CIN8, K_TR2 = range(10)   # uig_debug_last_conv_kernel()

_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/uig.h declares (tests/test_abi.py checks this)
SIGNATURES = {
    "uig_version": (C.c_char_p, []),
    "uig_last_error": (C.c_char_p, []),
    "uig_device_ok": (_i, []),
    "uig_debug_last_conv_kernel": (_i, []),
    "uig_debug_set_tile": (None, [_i]),
    "uig_debug_set_strip": (None, [_i]),
    "uig_debug_set_strip_wide": (None, [_i]),
    "uig_debug_set_mirror": (None, [_i]),
    "uig_debug_set_strip_pk": (None, [_i, _i]),
    "uig_debug_strip_pk_phased_count": (C.c_long, []),
    "uig_debug_set_rowstrip": (None, [_i]),
    "uig_debug_set_strip_stages": (None, [_i]),
    "uig_debug_set_strip_small": (None, [_i]),
    "uig_debug_set_strip_small_stages": (None, [_i]),
    "uig_debug_set_infer_cs": (None, [_i]),
    "uig_debug_set_strip_stamps": (None, [_vp]),
    "uig_debug_set_mx_stamps": (None, [_vp]),
    "uig_debug_set_mx_issuers": (None, [_i]),
    "uig_conv2d_fwd_workspace_bytes": (_sz, [_i] * 10),
    "uig_conv2d_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp]),
    "uig_conv_transpose2d_fwd_workspace_bytes": (_sz, [_i] * 6),
    "uig_conv_transpose2d_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 6 + [_vp, _sz, _vp]),
    "uig_conv2d_bwd_workspace_bytes": (_sz, [_i] * 10),
    "uig_conv2d_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp]),
    "uig_conv_gather": (_i, [_vp, _vp, _vp, _vp] + [_i] * 15 + [_i, _f, _i, _vp]),
    "uig_conv_gather_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp] + [_i] * 15 + [_i, _f, _i, _vp]),
    "uig_conv_gather_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp] + [_i] * 15 + [_i, _f, _i, _vp]),
    "uig_conv_gather_bst": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp] + [_i] * 15 + [_i, _f, _i] + [_vp, _vp, _i, _f, _vp] + [_vp]),
    "uig_debug_set_in_tickets": (None, [_i]),
    "uig_debug_set_in_fused": (None, [_i]),
    "uig_instnorm_bwd_fused_applicable": (_i, [_i, _i64, _i, _i]),
    "uig_instnorm_act_bwd_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_debug_set_colsum_slabs": (None, [_i]),
    "uig_conv_gather_fin": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp] + [_i] * 15 + [_i, _f, _i] + [_vp, _f, _vp] + [_vp]),
    "uig_reflect3x3_dgrad_mirror_bst": (_i, [_vp, _vp, _vp, _i, _vp, _vp] + [_i] * 7 + [_vp, _vp, _i, _f, _vp, _vp, _vp] + [_vp]),
    "uig_instnorm_apply_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_instnorm_act_fwd_t": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "uig_instnorm_act_bwd_colsum_t": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_reflect3x3_dgrad_border": (_i, [_vp, _vp, _vp, _i, _vp] + [_i] * 7 + [_vp]),
    "uig_reflect3x3_dgrad_mirror_applicable": (_i, [_i] * 7),
    "uig_reflect3x3_dgrad_mirror": (_i, [_vp, _vp, _vp, _i, _vp, _vp] + [_i] * 7 + [_vp]),
    "uig_conv_strip_applicable": (_i, [_i] * 10),
    "uig_conv_strip_tile": (_i, [_i] * 10),
    "uig_conv3x3_innorm_applicable": (_i, [_i] * 8),
    "uig_conv3x3_innorm_fwd": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp] + [_i] * 7 + [_i, _f, _i, _vp]),
    "uig_debug_set_normconv": (None, [_i]),
    "uig_instnorm_finalize": (_i, [_vp, _i, _vp, _i, _i64, _i, _f, _vp]),
    "uig_conv3x3_mx_fp8_applicable": (_i, [_i] * 5),
    "uig_conv3x3_mx_fp8": (_i, [_vp] * 8 + [_i] + [_vp] * 4 + [_i] * 8 + [_i, _f] + [_vp, _vp, _i, _f, _vp] + [_vp]),
    "uig_mx_quantize": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "uig_mx_quantize_multi": (_i, [_vp, _i, _i64, _vp]),
    "uig_conv3x3_mx_fp8_dgrad_mirror_applicable": (_i, [_i] * 5),
    "uig_conv3x3_mx_fp8_dgrad_mirror": (_i, [_vp] * 6 + [_i, _vp, _vp] + [_i] * 6 + [_vp]),
    "uig_instnorm_act_fwd_mx": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "uig_instnorm_act_bwd_colsum_pre": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_instnorm_act_bwd_colsum_mx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "uig_wgrad_tile_rows": (_i, [_i, _i, _i]),
    "uig_debug_set_wgrad_wide": (None, [_i]),
    "uig_debug_set_wgrad_rows": (None, [_i]),
    "uig_debug_set_wgrad_rows_s2": (None, [_i]),
    "uig_debug_set_wgrad_head": (None, [_i]),
    "uig_debug_set_gemv": (None, [_i]),
    "uig_debug_set_cin8": (None, [_i]),
    "uig_debug_set_tr2": (None, [_i]),
    "uig_conv_tr2_applicable": (_i, [_i] * 8),
    "uig_wgrad_splits": (_i, [_i] * 13),
    "uig_wgrad_partial": (_i, [_vp, _vp, _vp] + [_i] * 14 + [_vp]),
    "uig_wgrad_pair_splits": (_i, [_i] * 14),
    "uig_wgrad_partial_pair": (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "uig_wgrad_pair2_splits": (_i, [_i] * 16),
    "uig_wgrad_partial_pair2": (_i, [_vp] * 5 + [_i] * 18 + [_vp]),
    "uig_wgrad_reduce_pair": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "uig_wgrad_reduce_pair2": (_i, [_vp] * 3 + [_i] * 7 + [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "uig_wgrad_reduce": (_i, [_vp, _vp] + [_i] * 7 + [_vp]),
    "uig_wgrad_reduce_bias": (_i, [_vp, _vp] + [_i] * 7 + [_vp, _i, _i, _i, _vp, _i, _vp]),
    "uig_colsum_workspace_floats": (_sz, [_i]),
    "uig_bias_grad": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "uig_pack_weight": (_i, [_vp, _vp] + [_i] * 9 + [_vp]),
    "uig_pack_tiles": (_i, [_i] * 7),
    "uig_pack_weights_multi": (_i, [_vp, _i, _i64, _i, _vp]),
    "uig_pack_tiles2": (_i, [_i] * 7),
    "uig_pack_weights_multi2": (_i, [_vp, _i, _i64, _i, _vp]),
    "uig_instnorm_workspace_floats": (_sz, [_i, _i64, _i]),
    "uig_instnorm_act_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "uig_instnorm_act_fwd_pre": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "uig_instnorm_act_fwd_infer": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _f, _i, _f, _i, _vp]),
    "uig_instnorm_act_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_instnorm_bwd_colsum_slabs": (_i, [_i, _i64, _i, _i]),
    "uig_instnorm_act_bwd_colsum": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp]),
    "uig_bias_grad_from_partials": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "uig_reflect_fold": (_i, [_vp, _vp] + [_i] * 6 + [_vp]),
    "uig_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "uig_l1_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f, _i, _vp]),
    "uig_mse_const_fwd_bwd": (_i, [_vp, _f, _vp, _vp, _vp, _i64, _f, _i, _vp]),
    "uig_loss_workspace_floats": (_sz, []),
    "uig_scale_by_scalar": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "uig_adam_flat": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp]),
    "uig_adam_flat_graph": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _vp, _f, _vp]),
    "uig_to_nhwc": (_i, [_vp, _i, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "uig_from_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i64, _i64, _i64, _i64, _vp]),
    "uig_resize_crop_flip_normalize": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libuig.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C unpaired-image-generation_amd/csrc`). There is no fallback path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        # A/B switches without rebuilding: UIG_DEBUG_HOOKS="gemv=0,wgrad_rows=0" calls uig_debug_set_<name>(value)
        for item in filter(None, os.environ.get("UIG_DEBUG_HOOKS", "").split(",")):
            name, _, val = item.partition("=")
            getattr(l, "uig_debug_set_" + name.strip())(*[int(v) for v in val.split(":")])      # "strip_pk=2:0": two-argument hook
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (status {rc}): {lib().uig_last_error().decode(errors='replace')}")
