"""ctypes binding of libfocoos_amd.so (include/focoos_amd.h).

The HIP library is the product: if it is missing or a call fails we raise — there
is no PyTorch/CPU fallback anywhere in the package."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import LIB_PATH, LIB_PATH_FP16

FX_ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "gelu": 3}


class FocoosAmdError(RuntimeError):
    pass


class FxConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("N", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("pool2", C.c_int32), ("act", C.c_int32), ("out_f32", C.c_int32), ("residual_after_act", C.c_int32),
        ("y_batch_stride", C.c_int64), ("w_frag", C.c_void_p), ("mask", C.c_void_p), ("ldm", C.c_int32), ("reserved0", C.c_int32),
    ]


class FxPackEntry(C.Structure):
    """include/focoos_amd.h fx_pack_entry"""
    _fields_ = [
        ("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
        ("w_fwd_frag", C.c_void_p), ("w_dgrad_frag", C.c_void_p), ("bias_out", C.c_void_p),
        ("N", C.c_int32), ("C", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("ld_fwd", C.c_int32), ("ld_dgrad", C.c_int32),
        ("first_block", C.c_int32), ("n_offset", C.c_int32), ("n_total", C.c_int32), ("reserved", C.c_int32),
    ]


class FxPwChainDesc(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("residual", C.c_void_p), ("w1", C.c_void_p), ("bias1", C.c_void_p), ("y1", C.c_void_p),
        ("w2", C.c_void_p), ("bias2", C.c_void_p), ("y2", C.c_void_p),
        ("M", C.c_int32), ("K1a", C.c_int32), ("K1b", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32),
        ("ldx1", C.c_int32), ("ldx2", C.c_int32), ("ldr", C.c_int32), ("ldy1", C.c_int32), ("ldy2", C.c_int32),
        ("act1", C.c_int32), ("act2", C.c_int32),
        ("pool", C.c_void_p), ("ldp", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32), ("reserved0", C.c_int32),
    ]


class FxRcStage(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32),
        ("src", C.c_int32), ("dst", C.c_int32), ("aux", C.c_int32),
        ("ld", C.c_int32), ("ld2", C.c_int32), ("flags", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("g0", C.c_void_p), ("g1", C.c_void_p),
    ]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float

# name -> argtypes (every function returns int unless noted); mirrors include/focoos_amd.h
SIGNATURES = {
    "fx_abi_version": [],
    "fx_build_flags": [],
    "fx_device_info": [_i, C.POINTER(C.c_int), C.c_char_p, _i],
    "fx_conv2d_nhwc_bf16": [C.POINTER(FxConvDesc), _vp],
    "fx_conv2d_variant": [_vp, C.c_char_p, _i],
    "fx_pw_chain_supported": [_i, _i, _i, _i],
    "fx_pw_chain_pool_supported": [_i, _i, _i, _i],
    "fx_conv3x3_flat_supported": [_i, _i, _i],
    "fx_pw_chain_bf16": [C.POINTER(FxPwChainDesc), _vp],
    "fx_stem_conv3x3s2": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_resize_bilinear_u8": [_vp, _i, _i, _vp, _i, _i, _vp],
    "fx_maxpool3x3s2_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_stem_conv12_u8_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_stem_conv_pool_supported": [_i, _i, _i, _i],
    "fx_stem_conv3x3_relu_maxpool_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_avgpool2x2_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_resize_bilinear_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_add_rows_bf16": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "fx_layernorm_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "fx_mha_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_mha_workspace_bytes": [_i, _i, _i, _i, _i],
    "fx_mha_masked_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, C.c_size_t, _vp],
    "fx_upsample_nearest_add_nhwc_bf16": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_upsample_nearest_bwd_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_query_pixel_logits_bf16": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_mf_class_head": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "fx_mf_upsample_probs_f32": [_vp, _i, _i, _vp, _i, _i, _i, _vp],
    "fx_mf_upsample_probs_bf16": [_vp, _i, _i, _vp, _i, _i, _i, _vp],
    "fx_mf_postprocess_workspace_bytes": [_i, _i, _i],
    "fx_mf_postprocess_workspace_bytes_fused": [_i, _i, _i, _i, _i, _i],
    "fx_mf_postprocess": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _i, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fx_msda_bf16": [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_rowmax_f32": [_vp, _i, _vp, _i, _i, _vp],
    "fx_enc_score_head_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _vp, _i, _vp],
    "fx_row_chain": [_vp, _i, _i, _i, _vp],
    "fx_topk_rows_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "fx_topk_rows_workspace_bytes": [_i, _i, _i],
    "fx_topk_rows_ws_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, C.c_size_t, _vp],
    "fx_gather_rows_bf16": [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp],
    "fx_fill_rows_bf16": [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp],
    "fx_linear_k4_relu": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "fx_bbox_head": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "fx_detr_head_out": [_vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "fx_detr_postprocess": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "fx_detr_match_cost_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp],
    "fx_lsa_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "fx_lsa_status_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "fx_detr_set_loss_workspace_bytes": [_i, _i, _i],
    "fx_detr_set_loss_f32": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp],
    "fx_detr_box_loss_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "fx_detr_box_loss_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "fx_msda_f32_fwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_msda_f32_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_msda_train_fwd": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_msda_train_bwd": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_msda_bwd_slab_supported": [_vp, _i, _i, _i, _i, _i],
    "fx_msda_prep_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_msda_prep_bwd_bf16": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_msda_train_bwd_slab": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_adamw_workspace_bytes": [],
    "fx_adamw_step_f32": [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp],
    "fx_adamw_step_scaled_f32": [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp, _vp, _vp, _f, _f, _i, _vp],
    "fx_conv2d_wgrad_nhwc_bf16": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_conv2d_wgrad_bias_nhwc_bf16": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_point_sample_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "fx_mask_match_cost_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, _i, _vp, _vp],
    "fx_mask_match_cost_workspace_bytes": [_i, _i, _i],
    "fx_mask_match_cost_ws_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, _i, _vp, _vp, C.c_size_t, _vp],
    "fx_mask_set_loss_workspace_bytes": [_i, _i, _i, _i],
    "fx_mask_set_loss_f32": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float,
                             C.c_float, C.c_float, C.c_float, _vp, C.c_size_t, _vp, _vp],
    "fx_dwconv3x3s2_nhwc_bf16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_dwconv3x3s2_nhwc_f32out": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_global_mean_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _vp],
    "fx_pooled_linear_f32": [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "fx_channel_gate_nhwc_bf16": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "fx_seg_postprocess_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "fx_seg_postprocess": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, C.c_float, _i, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "fx_pack_conv_weights_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_unpack_conv_wgrad_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "fx_linear_wgrad_bias_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_conv2d_wgrad_splits": [_i, _i, _i, _i, _i, _i, _i],
    "fx_conv2d_wgrad_variant": [_i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_char_p, _i],
    "fx_conv2d_wgrad_partial_nhwc_bf16": [_vp, _i, _vp, _i, _vp, C.c_int64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_unpack_conv_wgrad_sum_f32": [_vp, C.c_int64, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "fx_relu_bwd_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, C.c_int64, _i, _i, _vp],
    "fx_zero_insert2_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_avgpool2x2_bwd_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_maxpool3x3s2_bwd_nhwc_bf16": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "fx_normalize_pad8": [_vp, _i, _vp, _vp, _vp, C.c_int64, _vp],
    "fx_stem_conv3x3s2_linear": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_bn_stats_bf16": [_vp, _i, _i, _vp, C.c_int64, _i, _vp],
    "fx_bn_finalize_f32": [_vp, C.c_float, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "fx_bn_apply_bf16": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, C.c_int64, _i, _vp],
    "fx_bn_bwd_stats_bf16": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, C.c_int64, _i, _vp],
    "fx_bn_bwd_apply_bf16": [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, C.c_float, _vp, _i, _vp, _i, C.c_int64, _i, _vp, _vp, _vp],
    "fx_act_fwd_bf16": [_vp, _i, _vp, _i, C.c_int64, _i, _i, _vp],
    "fx_act_bwd_bf16": [_vp, _i, _vp, _i, _vp, _i, C.c_int64, _i, _i, _vp],
    "fx_colsum_bf16": [_vp, _i, _vp, C.c_int64, _i, _vp],
    "fx_layernorm_bwd_bf16": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp],
    "fx_resize_bilinear_bwd_nhwc_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "fx_cast_f32_bf16": [_vp, _vp, C.c_int64, _vp],
    "fx_mha_bwd_workspace_bytes": [_i, _i, _i, _i],
    "fx_mha_bwd_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, C.c_size_t, _vp],
    "fx_mha_masked_bwd_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, C.c_size_t, _vp],
    "fx_mask_set_loss_bwd_f32": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, _vp, C.c_size_t, _vp, _vp, _i, _vp, _vp],
    "fx_planes_to_rows_bf16": [_vp, _i, _i, _vp, _i, _i, _i, _vp],
    "fx_dwconv3x3s2_bwd_nhwc_bf16": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "fx_rowdot_nhwc_bf16": [_vp, _i, _vp, _i, C.c_float, _vp, _i, _i, _i, _i, _i, _vp],
    "fx_bcast_vec_nhwc_bf16": [_vp, _i, C.c_float, _vp, _i, _i, _i, _i, _vp],
    "fx_scatter_rows_bf16": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "fx_pack_entry_blocks": [_i, _i, _i, _i],
    "fx_pack_weights_many_f32": [_vp, _i, _i, _vp],
    "fx_pack_frag_bf16": [_vp, _vp, _i, _i, _vp],
    "fx_pack_linear_weights_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "fx_box_refine_f32": [_vp, _vp, _vp, C.c_int64, _f, _vp],
    "fx_box_refine_bwd_f32": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _f, _vp],
    "fx_vfl_loss_bf16": [_vp, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _i, C.c_int64, _i, _vp],
    "fx_stream_fork": [_vp, _vp],
    "fx_stream_join": [_vp, _vp],
    "fx_graph_begin": [_vp],
    "fx_graph_end": [_vp, C.POINTER(C.c_void_p)],
    "fx_graph_launch": [_vp, _vp],
    "fx_graph_destroy": [_vp],
    "fx_graph_time": [_vp, _vp, _i, C.POINTER(C.c_float)],
}

# One library per 16-bit storage element: "bf16" = libfocoos_amd.so (the product: inference engines and the default training step),
# "fp16" = libfocoos_amd_fp16.so (same sources, -DFX_FP16=1: the training step under a loss scale - BASELINE configs[4], the reference's
# fp16 autocast + GradScaler).  The CURRENT element type is process state: everything that allocates activations / weight images asks
# act_dtype(), everything that launches asks load(); a TrainStep pins its own type at the top of every step.  Inference engines are bf16.
_libs: dict = {}
_DTYPE = [os.environ.get("FX_DTYPE", "bf16")]


def compute_dtype() -> str:
    return _DTYPE[0]


def set_compute_dtype(name: str) -> str:
    """Select the library / element type used by objects created and steps run from now on; returns the previous one."""
    if name not in ("bf16", "fp16"):
        raise FocoosAmdError(f"compute dtype {name!r}: the engine has a bfloat16 and an fp16 (+ loss scale) build")
    prev, _DTYPE[0] = _DTYPE[0], name
    return prev


class using_compute_dtype:
    """Scoped ``set_compute_dtype``: the previous element type is restored on exit, exception or not (a step or a training run
    that pins its type must not leave the process in it - a later lazy ``load()`` would pick the other library)."""

    def __init__(self, name: str):
        self.name, self.prev = name, None

    def __enter__(self):
        self.prev = set_compute_dtype(self.name)
        return self

    def __exit__(self, *exc):
        _DTYPE[0] = self.prev
        return False


def act_dtype():
    """torch dtype of activations, packed weight images and activation gradients under the current element type."""
    import torch

    return torch.float16 if _DTYPE[0] == "fp16" else torch.bfloat16


def lib_path(name: str = None) -> str:
    if (name or _DTYPE[0]) == "fp16":
        return os.environ.get("FOCOOS_AMD_LIB_FP16", LIB_PATH_FP16)
    return os.environ.get("FOCOOS_AMD_LIB", LIB_PATH)


FX_ABI_VERSION = 8   # = include/focoos_amd.h (tests/test_host_cpu.py compares the two)


def load(name: str = None) -> C.CDLL:
    """Load the HIP library of the current element type (or of ``name``: the inference-side processors always bind the bf16 product library,
    whatever type a training step of the same process pinned); raise loudly when it is absent (no fallback)."""
    name = name or _DTYPE[0]
    if name in _libs:
        return _libs[name]
    import torch  # noqa: F401  -- must be imported first: it loads the HIP runtime (its bundled libamdhip64) our .so binds to
    path = lib_path(name)
    if not os.path.exists(path):
        raise FocoosAmdError(
            f"{path} not found: the gfx950 HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m focoos_amd.build`). focoos_amd has no CPU/PyTorch fallback by design."
        )
    lib = C.CDLL(path)
    for sym, argtypes in SIGNATURES.items():
        fn = getattr(lib, sym)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.fx_mha_bwd_workspace_bytes.restype = C.c_size_t
    lib.fx_seg_postprocess_workspace_bytes.restype = C.c_size_t
    lib.fx_mask_match_cost_workspace_bytes.restype = C.c_size_t
    lib.fx_mf_postprocess_workspace_bytes_fused.restype = C.c_size_t
    lib.fx_mask_set_loss_workspace_bytes.restype = C.c_size_t
    lib.fx_topk_rows_workspace_bytes.restype = C.c_size_t
    lib.fx_error_string.argtypes = [C.c_int]
    lib.fx_error_string.restype = C.c_char_p
    if lib.fx_abi_version() != FX_ABI_VERSION:
        raise FocoosAmdError(f"ABI version mismatch: library {lib.fx_abi_version()} != binding {FX_ABI_VERSION} (stale {os.path.basename(path)}: rebuild)")
    if bool(lib.fx_build_flags() & 2) != (name == "fp16"):
        raise FocoosAmdError(f"{path} was built for the other 16-bit element type (fx_build_flags = {lib.fx_build_flags()}) than {name!r}")
    _libs[name] = lib
    return lib


_warned_pk = False


def two_queue_safe() -> bool:
    """False when the loaded library contains packed-fp32 code (fx_build_flags bit 0: FX_PK_F32 builds - the reproducer of the two-queue
    hazard): callers then keep everything on ONE hardware queue (one batch part, no weight-gradient side stream).
    FX_ALLOW_PK_TWO_QUEUES=1 overrides (the reproducer itself: tests/test_gpu_two_streams.py, scripts/dev/pk_bisect.sh)."""
    global _warned_pk
    if not (load().fx_build_flags() & 1) or os.environ.get("FX_ALLOW_PK_TWO_QUEUES") == "1":
        return True
    if not _warned_pk:
        import warnings

        warnings.warn("libfocoos_amd was built with packed-fp32 instructions (FX_PK_F32): concurrent batch parts and the weight-gradient "
                      "side stream are disabled (two-queue hazard); rebuild without FX_PK_F32 for the default configuration")
        _warned_pk = True
    return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().fx_error_string(rc).decode()
        raise FocoosAmdError(f"libfocoos_amd {what} failed: {msg} (code {rc})")
