"""ctypes binding of libdgx.so (C ABI: include/divergen_hip.h).

There is NO fallback: if the library is missing or an op is called on a non-GPU tensor the call
raises.  The CPU oracle lives under oracle/ and is never imported from this package.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGX_LIB", os.path.join(_HERE, "csrc", "libdgx.so"))  # DGX_LIB: dev A/B builds
_lib = None

c_p, c_i, c_f, c_i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64


class WgradProblem(ctypes.Structure):
    """struct dgx_wgrad_problem (include/divergen_hip.h)."""
    _fields_ = [("dy", c_p), ("x", c_p), ("gw", c_p), ("M", c_i), ("Nn", c_i), ("Kk", c_i), ("gb", c_p)]


class GemmEpilogue(ctypes.Structure):
    """struct dgx_gemm_epilogue (include/divergen_hip.h)."""
    _fields_ = [("mode", c_i), ("c", c_p), ("ldc", c_i64), ("bias", c_p), ("c2", c_p), ("aux", c_p), ("ldaux", c_i64),
                ("residual", c_p), ("out", c_p), ("scale", c_p), ("residual_dtype", c_i),
                ("B", c_i), ("H", c_i), ("W", c_i), ("ws", c_i), ("shift", c_i), ("workspace", c_p), ("workspace_bytes", c_i64), ("relu", c_i)]


class ProfStats(ctypes.Structure):
    """struct dgx_prof_stats (include/divergen_hip.h)."""
    _fields_ = [("ms", ctypes.c_double), ("launches", c_i64), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("captured_launches", c_i64), ("captured_flops", ctypes.c_double), ("captured_bytes", ctypes.c_double)]


class PadItem(ctypes.Structure):
    """struct dgx_pad_item (include/divergen_hip.h)."""
    _fields_ = [("x", c_p), ("xpad", c_p), ("N", c_i), ("H", c_i), ("W", c_i)]


class ConvItem(ctypes.Structure):
    """struct dgx_conv_item (include/divergen_hip.h)."""
    _fields_ = [("xpad", c_p), ("y", c_p), ("N", c_i), ("H", c_i), ("W", c_i)]


class ConvWgradItem(ctypes.Structure):
    """struct dgx_conv_wgrad_item (include/divergen_hip.h)."""
    _fields_ = [("dypad", c_p), ("xpad", c_p), ("N", c_i), ("H", c_i), ("W", c_i)]


class HeadLevel(ctypes.Structure):
    """struct dgx_head_level (include/divergen_hip.h)."""
    _fields_ = [("x", c_p), ("dx", c_p), ("scale", c_p), ("rows", c_i)]


class GnItem(ctypes.Structure):
    """struct dgx_gn_item (include/divergen_hip.h)."""
    _fields_ = [("x", c_p), ("dy", c_p), ("out", c_p), ("mean", c_p), ("rstd", c_p), ("scratch", c_p), ("N", c_i), ("HW", c_i)]


class ColsumProblem(ctypes.Structure):
    """struct dgx_colsum_problem (include/divergen_hip.h)."""
    _fields_ = [("dy", c_p), ("out", c_p), ("M", c_i), ("N", c_i)]


# name -> (restype, argtypes); must list every symbol include/divergen_hip.h declares
SIGNATURES = {
    "dgx_build_arch": (ctypes.c_char_p, []),
    "dgx_abi_version": (c_i, []),
    "dgx_host_register": (c_i, [c_p, ctypes.c_size_t]),
    "dgx_host_unregister": (c_i, [c_p]),
    "dgx_memcpy_h2d_async": (c_i, [c_p, c_p, ctypes.c_size_t, c_p]),
    "dgx_window_attention_fwd": (c_i, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dgx_window_attention_bwd": (c_i, [c_p] * 8 + [c_i64, c_i64, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dgx_window_attention_fwd_compact": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dgx_window_attention_bwd_compact": (c_i, [c_p] * 9 + [c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dgx_window_gather": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_window_scatter": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_roi_align_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_roi_align_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_roi_pooler_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_roi_pooler_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_preprocess_patches": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "dgx_stem_im2col7x7": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "dgx_affine_act_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "dgx_affine_act_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p]),
    "dgx_maxpool3x3s2_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_maxpool3x3s2_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_mask_bce_workspace_floats": (c_i64, [c_i64]),
    "dgx_mask_bce": (c_i, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i, c_p]),
    "dgx_prof_enable": (c_i, [c_i]),
    "dgx_prof_pause": (c_i, [c_i]),
    "dgx_prof_read": (c_i, [c_i, c_p]),
    "dgx_roi_pooler_bwd_gather": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_roi_pooler_bwd_gather_accum": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dgx_mask_crop": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_nms_sorted": (c_i, [c_p, c_i, c_f, c_p, c_p, c_p, c_p]),
    "dgx_nms_workspace_words": (c_i64, [c_i]),
    "dgx_nms_batched_workspace_words": (c_i64, [c_i, c_i]),
    "dgx_nms_batched": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_p, c_i, c_p, c_p]),
    "dgx_iou_match": (c_i, [c_p, c_i, c_p, c_i, c_f, c_p, c_p, c_p, c_p]),
    "dgx_centernet_targets": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p, c_p]),
    "dgx_centernet_scores": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_i, c_p]),
    "dgx_centernet_decode": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_f, c_p, c_p, c_p, c_i, c_p]),
    "dgx_centernet_head_outputs": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "dgx_centernet_head_outputs_bwd_workspace_floats": (c_i64, [c_p, c_i, c_i]),
    "dgx_centernet_head_outputs_bwd": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "dgx_set_reserved_cus": (None, [c_i]),
    "dgx_get_reserved_cus": (c_i, []),
    "dgx_dev_set": (c_i, [ctypes.c_char_p, c_i]),
    "dgx_gemm_last_form": (c_i, [ctypes.POINTER(c_i)] * 3),
    "dgx_dev_gemm_log": (c_i, [ctypes.c_char_p]),
    "dgx_gather_boxes": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "dgx_topk_index_rows": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_i64, c_p]),
    "dgx_sort_rows_desc": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "dgx_centernet_finalize": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "dgx_roi_label": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "dgx_roi_gather": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f] + [c_p] * 10 + [c_p]),
    "dgx_centernet_label_inds": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "dgx_copy_paste": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "dgx_im2col3x3": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_col2im3x3": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_wgrad_workspace_bytes": (c_i64, [c_i, c_i, c_i]),
    "dgx_linear_wgrad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "dgx_layernorm_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_patch_merge_ln_fwd": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dgx_patch_merge_ln_bwd": (c_i, [c_p] * 9 + [c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_layernorm_f32out_fwd": (c_i, [c_p] * 6 + [c_i64, c_i, c_f, c_i, c_p]),
    "dgx_layernorm_f32out_bwd": (c_i, [c_p] * 9 + [c_i64, c_i, c_i, c_p]),
    "dgx_layernorm_bwd_blocks": (c_i, [c_i64]),
    "dgx_layernorm_bwd": (c_i, [c_p] * 10 + [c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_layernorm_bwd_emit": (c_i, [c_p] * 10 + [c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_layernorm_param_reduce2": (c_i, [c_p] * 6 + [c_i64, c_i, c_p]),
    "dgx_layernorm_param_reduce_n": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_i, c_p]),
    "dgx_wgrad_grouped_workspace_bytes": (c_i64, [ctypes.POINTER(WgradProblem), c_i]),
    "dgx_linear_wgrad_grouped": (c_i, [ctypes.POINTER(WgradProblem), c_i, c_f, c_p, c_p]),
    "dgx_wgrad_grouped_form": (c_i, [ctypes.POINTER(WgradProblem), c_i]),
    "dgx_cascade_refine": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_f, c_f, c_f, c_f, c_f,
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "dgx_paste_masks": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dgx_paste_rle": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dgx_rle_encode": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_rle_to_string": (c_i64, [c_p, c_i64, c_p, c_i64]),
    "dgx_grad_bank_update": (c_i, [c_p, c_p, c_i64, c_f, c_f, c_p]),
    "dgx_grad_sim_workspace_bytes": (c_i64, [c_i64]),
    "dgx_grad_sim": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "dgx_centernet_losses_blocks": (c_i, [c_i]),
    "dgx_centernet_losses": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_i] + [c_f] * 6 + [c_p] * 5 + [c_p]),
    "dgx_detic_losses": (c_i, [c_p] * 7 + [c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p]),
    "dgx_detic_losses_strided": (c_i, [c_p, c_i64, c_p, c_i64] + [c_p] * 5 + [c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i64, c_i, c_p, c_p, c_p, c_i, c_p]),
    "dgx_detic_grad_scale": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "dgx_fed_class_mask": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "dgx_gelu_fwd": (c_i, [c_p, c_p, c_i64, c_p]),
    "dgx_gelu_bwd_workspace_bytes": (c_i64, [c_i, c_i]),
    "dgx_gelu_bwd_colsum": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p]),
    "dgx_groupnorm_scratch_floats": (c_i64, [c_i, c_i, c_i]),
    "dgx_groupnorm_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dgx_groupnorm_bwd": (c_i, [c_p] * 10 + [c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_groupnorm_fwd_multi": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_f, c_i, c_p]),
    "dgx_groupnorm_bwd_multi": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "dgx_colsum_workspace_bytes": (c_i64, [c_i, c_i]),
    "dgx_colsum_bf16": (c_i, [c_p, c_p, c_i, c_i, c_f, c_p, c_p]),
    "dgx_colsum_grouped_workspace_bytes": (c_i64, [ctypes.POINTER(ColsumProblem), c_i]),
    "dgx_colsum_grouped": (c_i, [ctypes.POINTER(ColsumProblem), c_i, c_f, c_p, c_p]),
    "dgx_residual_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_residual_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_upsample2x_add_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_upsample2x_add_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dgx_gemm_bf16_nt": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i64, c_i64, ctypes.POINTER(GemmEpilogue), c_p]),
    "dgx_conv3x3_pad_rows": (c_i64, [c_i, c_i, c_i]),
    "dgx_conv3x3_pad": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_conv3x3_pad_relu_grad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_conv3x3_pad_multi": (c_i, [c_p, c_i, c_i, c_p]),
    "dgx_deconv2x2_shuffle": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_deconv2x2_unshuffle_relu_grad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dgx_conv3x3_gemm_multi": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p]),
    "dgx_conv3x3_wgrad_bias_multi_workspace_bytes": (c_i64, [c_p, c_i, c_i, c_i]),
    "dgx_conv3x3_wgrad_bias_multi": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_p]),
    "dgx_conv3x3_gemm": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p]),
    "dgx_conv3x3_wgrad_workspace_bytes": (c_i64, [c_i, c_i, c_i, c_i, c_i]),
    "dgx_conv3x3_wgrad": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "dgx_conv3x3_wgrad_bias_workspace_bytes": (c_i64, [c_i, c_i, c_i, c_i, c_i]),
    "dgx_conv3x3_wgrad_bias": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "dgx_transpose_bf16_grouped": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_p]),
    "dgx_zero_ranges_f32": (c_i, [c_p, c_p, c_i64, c_p]),
    "dgx_sgd_ema_step": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_i, c_f, c_f, c_f, c_p, c_i, c_f, c_p, c_p, c_i, c_p, c_p]),
    "dgx_clip_coef_workspace_floats": (c_i64, []),
    "dgx_clip_coef_f32": (c_i, [c_p, c_i64, c_f, c_f, c_p, c_p, c_p]),
    "dgx_adamw_ema_step": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f,
                                 c_p, c_p, c_i, c_p, c_p]),
    "dgx_adamw_ema_step_scaled": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_i, c_f,
                                        c_p, c_p, c_i, c_p, c_p]),
}


class DgxError(RuntimeError):
    pass


def lib():
    """Load libdgx.so once.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DgxError("libdgx.so is missing (%s): run `python -m divergen_amd.csrc.build` or "
                           "__graft_entry__.build(); there is no CPU/eager fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise DgxError("%s failed with code %d" % (what, code))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Must be a contiguous GPU tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DgxError("libdgx ops need GPU (ROCm) tensors; got a %s tensor -- no CPU fallback exists" % t.device)
    if not t.is_contiguous():
        raise DgxError("libdgx ops need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


_RESERVED = [0]


def set_reserved_cus(n):
    """dgx_set_reserved_cus with the value remembered on this side (utils/graphs.py restores it around a capture)."""
    lib().dgx_set_reserved_cus(int(n))
    _RESERVED[0] = int(n)


def reserved_cus():
    return _RESERVED[0]


def stream():
    """hipStream_t of torch's current stream on the current device.  `torch.cuda.current_stream().cuda_stream` builds a Stream
    object through five layers of Python (device-index resolution, availability probes, an os.environ lookup): 9 us per call,
    ~180 calls = 1.9 ms of host time per training step (tools/host_profile.py); the raw accessors take ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


DGX_F32, DGX_BF16 = 0, 1


def dtype_code(t):
    if t.dtype == torch.float32:
        return DGX_F32
    if t.dtype == torch.bfloat16:
        return DGX_BF16
    raise DgxError("unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)
