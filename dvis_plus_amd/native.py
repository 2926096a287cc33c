"""ctypes binding of libdvis_hip.so — the C ABI declared in include/dvis_hip.h.

This is the ONLY way product code reaches the kernels.  There is no CPU or eager fallback: if the
library is missing, or an entry point is called on non-GPU memory, a ``RuntimeError`` is raised
(the reference instead swallows every failure, ops/modules/ms_deform_attn.py:116-121).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVIS_HIP_LIB") or os.path.join(_HERE, "lib", "libdvis_hip.so")      # (DVIS_HIP_LIB: development — A / B of two builds)

F32, F64, F16, BF16 = 0, 1, 2, 3
_DTYPE = {torch.float32: F32, torch.float64: F64, torch.float16: F16, torch.bfloat16: BF16}

_c = ctypes
_i, _i64, _p, _f = _c.c_int, _c.c_int64, _c.c_void_p, _c.c_float

# name -> (restype, argtypes); mirrors include/dvis_hip.h one-to-one (checked by tests/test_host_cpu.py::test_library_exports_every_declared_symbol)
SIGNATURES = {
    "dvis_last_error": (_c.c_char_p, []),
    "dvis_version": (_i, []),
    "dvis_msda_forward": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dvis_msda_backward": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dvis_msda_backward_det_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "dvis_msda_backward_det": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dvis_msda_fused_forward": (_i, [_p, _p, _p, _p, _i, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dvis_msda_fused_forward_pos": (_i, [_p, _p, _p, _p, _i, _p, _i64, _p, _i64, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i,
                                         _p, _p, _p]),
    "dvis_msda_fused_forward_h": (_i, [_i, _p, _p, _p, _p, _i, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dvis_msda_fused_forward_slots": (_i, [_p, _p, _p, _p, _i, _p, _i64, _p, _i64, _i, _i, _i, _p, _p, _i64, _i, _i, _i, _i,
                                           _i, _i, _i, _p, _p, _p]),
    "dvis_nchw_to_tokens": (_i, [_p, _p, _i64, _i, _i64, _i64, _i64, _p]),
    "dvis_tokens_to_nchw": (_i, [_p, _p, _i64, _i, _i64, _i64, _i64, _p]),
    "dvis_normalize_pad": (_i, [_p, _i, _p, _i64, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dvis_nchw_to_tokens_affine": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i64, _i64, _i64, _p]),
    "dvis_mask_logits": (_i, [_p, _p, _i, _i, _i, _i64, _p, _p]),
    "dvis_attn_mask": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dvis_center_pool3": (_i, [_p, _i64, _i, _i, _p, _p, _p, _p]),
    "dvis_attn_mask_pooled": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dvis_attention_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "dvis_attention_ws_bytes_k": (_i64, [_i, _i, _i, _i, _i]),
    "dvis_attention_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "dvis_attention_forward_k": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _i]),
    "dvis_add_layernorm": (_i, [_p, _p, _i64, _p, _p, _p, _i64, _i, _f, _p]),
    "dvis_add_layernorm_pos": (_i, [_p, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i, _f, _p]),
    "dvis_bias_act": (_i, [_p, _p, _p, _i64, _i, _i64, _i, _p]),
    "dvis_conv3x3_winograd_supported": (_i, [_i, _i, _i, _i]),
    "dvis_conv3x3_winograd_pack": (_i, [_p, _p, _i, _i, _p]),
    "dvis_conv3x3_winograd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv3x3s2_supported": (_i, [_i, _i, _i, _i]),
    "dvis_conv3x3s2_pack": (_i, [_p, _p, _i, _i, _p]),
    "dvis_conv3x3s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv7x7s2_supported": (_i, [_i, _i, _i, _i]),
    "dvis_conv7x7s2_pack": (_i, [_p, _p, _p]),
    "dvis_conv7x7s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "dvis_conv1x1_mfma_supported": (_i, [_i, _i, _i64]),
    "dvis_conv1x1_mfma_pack": (_i, [_p, _p, _i, _i, _p]),
    "dvis_conv1x1_mfma": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p]),
    "dvis_conv1x1s2_mfma": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv1x1_supported": (_i, [_i, _i, _i64]),
    "dvis_conv1x1_bias_act": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p]),
    "dvis_bias_relu_maxpool": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p]),
    "dvis_group_norm_affine": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i64, _f, _p]),
    "dvis_scale_shift_act": (_i, [_p, _p, _p, _i64, _i64, _i, _p]),
    "dvis_upsample_add_affine": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "dvis_upsample_add": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "dvis_dwconv3x3_tokens": (_i, [_p, _p, _i64, _i, _i, _i, _i, _p, _p, _i, _p]),
    "dvis_adapter_res2": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "dvis_vps_argmax": (_i, [_p, _i64, _i64, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dvis_vss_argmax": (_i, [_p, _i64, _i64, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dvis_resize2_gt0": (_i, [_p, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dvis_resize2": (_i, [_p, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dvis_lsap_solve": (_i, [_p, _i, _i, _p]),
    "dvis_match_chain": (_i, [_p, _i, _i, _p]),
    "dvis_gemm_nt": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_gemm_nt_hm": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, _i, _i, _i,
                             _i64, _p]),
    "dvis_gemm_nt_bb": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, _i, _i,
                             _p]),
    "dvis_gemm_ln": (_i, [_p, _i64, _p, _i64, _p, _p, _f, _p, _p, _f, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _i, _i, _i,
                          _i, _i, _p]),
    "dvis_gemm_ln_supported": (_i, [_i, _i, _i, _i]),
    "dvis_gemm_ln_num_configs": (_i, []),
    "dvis_gemm_ln_pick_config": (_i, [_i, _i, _i]),
    "dvis_x3_packed_bytes": (_i64, [_i, _i]),
    "dvis_x3_set_reserve": (_i, [_i]),
    "dvis_x3_set_range_flag": (_i, [_p]),
    "dvis_x3_set_tag": (_i, [_i]),
    "dvis_x3_pack": (_i, [_p, _i64, _i, _i, _i, _p, _p]),
    "dvis_x3_linear_supported": (_i, [_i, _i, _i]),
    "dvis_x3_linear": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _i, _p, _i64, _p]),
    "dvis_x3_tile_supported": (_i, [_i, _i]),
    "dvis_x3_tile_packed_bytes": (_i64, [_i, _i]),
    "dvis_x3_tile_pack": (_i, [_p, _i64, _i, _i, _i, _p, _p]),
    "dvis_x3_tile_linear_qkv": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _i, _i, _f, _p, _p]),
    "dvis_attention_x3_packed": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "dvis_x3_rows_image_bytes": (_i64, [_i64, _i]),
    "dvis_x3_rows_image": (_i, [_p, _i64, _i64, _i, _i, _p, _p]),
    "dvis_layernorm_rows_image": (_i, [_p, _p, _p, _i64, _i, _f, _i, _p, _p]),
    "dvis_x3_tile_pack_order": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p]),
    "dvis_x3_tile_linear_image": (_i, [_p, _i64, _i, _p, _i, _i, _i, _p, _i, _p, _i64, _p, _i64, _p, _i, _p]),
    "dvis_x3_tile_linear_qkv_image": (_i, [_p, _i64, _i, _p, _i, _i, _i, _p, _i, _i, _f, _p, _p]),
    "dvis_attention_x3_packed_image": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dvis_x3_tile_linear": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _i, _p, _i64, _p, _i64, _p]),
    "dvis_x3_linear_res": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _i, _p, _i64, _p, _i64, _p]),
    "dvis_x3_linear_add": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _i64, _p, _i, _p, _i64, _p]),
    "dvis_x3_linear_ln": (_i, [_p, _i64, _i64, _i, _p, _i, _i, _i, _p, _p, _i64, _p, _p, _f, _p, _i64, _p, _p, _i64, _p]),
    "dvis_x3_ffn_packed_bytes": (_i64, [_i, _i, _i]),
    "dvis_x3_ffn_pack": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "dvis_x3_ffn_ln": (_i, [_p, _i64, _i64, _i, _i, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _i64, _p]),
    "dvis_conv1x1_x3_supported": (_i, [_i, _i, _i64, _i64, _i64]),
    "dvis_conv1x1_x3_packed_bytes": (_i64, [_i, _i]),
    "dvis_conv1x1_x3_pack": (_i, [_p, _i, _i, _i, _p, _p]),
    "dvis_conv1x1_x3": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv1x1_x3_dual": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv3x3_x3_packed_bytes": (_i64, [_i, _i]),
    "dvis_conv3x3_x3_pack": (_i, [_p, _i, _i, _i, _p, _p]),
    "dvis_conv3x3_x3": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_bneck_x3_supported": (_i, [_i64, _i, _i]),
    "dvis_bneck_x3_image_bytes": (_i64, [_i64, _i, _i]),
    "dvis_bneck_x3_packed_bytes": (_i64, [_i, _i]),
    "dvis_bneck_x3_pack": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "dvis_bneck_x3": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv1x1_x3_image": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_conv_x3_image_bytes": (_i64, [_i64, _i, _i, _i]),
    "dvis_conv_x3_pack_image": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "dvis_conv_x3_image": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_upsample_add_image": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dvis_gemm_num_configs": (_i, []),
    "dvis_gemm_pick_config": (_i, [_i, _i, _i, _i]),
    "dvis_gemm_pick_config_nw": (_i, [_i, _i, _i, _i, _i]),
    "dvis_gemm_config_waves": (_i, [_i]),
}

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m dvis_plus_amd.build` "
                "(__graft_entry__.build()).  dvis_plus_amd has no CPU/eager fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dvis_last_error().decode()
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dtype_code(t):
    try:
        return _DTYPE[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}") from None


def dev_ptr(t, name):
    """Device pointer of a contiguous GPU tensor; loud failure otherwise."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (dvis_plus_amd has no CPU path); got device {t.device}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
