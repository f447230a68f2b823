"""ctypes binding of libdpmn_hip.so (include/dpmn_hip.h).

The HIP library IS the product path: there is no CPU fallback.  Importing this module loads the
shared object and fails loudly if it is missing or lacks a declared symbol; calling any op with
non-CUDA tensors raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPMN_HIP_LIB: alternative build of the same library (A/B kernel experiments from tools/); never a fallback
LIB_PATH = os.environ.get("DPMN_HIP_LIB") or os.path.join(_HERE, "lib", "libdpmn_hip.so")

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        "dpmn_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C dpmn_amd/csrc`).  There is no CPU fallback for the DPMN hot path." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

fp = C.c_void_p  # device float*


class PgrmBlock(C.Structure):
    _fields_ = [(n, fp) for n in (
        "norm1_q_w", "norm1_q_b", "norm1_kv_w", "norm1_kv_b", "q_w", "q_b", "kv_w", "kv_b")] + [
        ("bias_table", fp * 4)] + [(n, fp) for n in (
            "sk_proj_w", "sk_proj_b", "sk_fc1_w", "sk_fc1_b", "sk_fc2_w", "sk_fc2_b", "sk_head_w", "sk_head_b",
            "norm2_w", "norm2_b", "fc1_w", "fc1_b", "dw_w", "dw_b", "pw_w", "pw_b", "fc2_w", "fc2_b")]


class PgrmWeights(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "img_h", "img_w", "patch", "dim", "n_groups", "heads_per_group", "mlp_hidden", "hidden_size",
        "n_weight_list")] + [("window", C.c_int * 4)] + [(n, fp) for n in (
            "prior_fusion_w", "prior_fusion_b", "pe_w", "pe_b", "pe_norm_w", "pe_norm_b")] + [
        ("blocks", PgrmBlock * 2)] + [(n, fp) for n in ("tail0_w", "tail0_b", "tail1_w", "tail1_b")] + [
        ("weight_list", fp * 16), ("reuse_folded", C.c_int)]


class PgrmSavedBlock(C.Structure):
    NAMES = ("cat", "fold", "feats", "partial", "avec", "x1", "ypre", "V", "n2", "gpre", "g", "z", "tkv_out")
    _fields_ = [(n, fp) for n in NAMES]


class PgrmSaved(C.Structure):
    _fields_ = [("tq", fp), ("tkv0", fp), ("blk", PgrmSavedBlock * 2), ("c0", fp), ("c1", fp)]


class PgrmDrop(C.Structure):
    _fields_ = [("p", C.c_float), ("pa", C.c_float), ("dp", C.c_float * 2), ("seeds", C.c_ulonglong * 12)]


class PgrmBlockT(C.Structure):
    NAMES = ("fc2_t", "pw_t", "fc1_t", "head_t", "proj_t", "q_t", "kv_t")
    _fields_ = [(n, fp) for n in NAMES]


class ConvDesc(C.Structure):
    _fields_ = [("inp", fp * 3), ("in_scale", fp * 3), ("in_shift", fp * 3), ("cseg", C.c_int * 3)] + [
        (n, C.c_int) for n in ("B", "Hin", "Win", "KH", "KW", "stride", "dil_y", "dil_x", "pad_y", "pad_x", "Hp", "Wp",
                               "Hout", "Wout", "ostep", "ooy", "oox", "pro_act")] + [
        ("w", fp), ("bias", fp), ("Cout", C.c_int), ("epi_act", C.c_int), ("slope", C.c_float), ("res", fp),
        ("out", fp), ("out_ld", C.c_int), ("out_coff", C.c_int), ("out_nchw", C.c_int), ("pixel_shuffle", C.c_int),
        ("stats", fp), ("splitk_ws", fp), ("splitk_ws_bytes", C.c_size_t), ("nphase", C.c_int), ("w_phase_stride", C.c_long),
        ("groups", C.c_int), ("w_group_stride", C.c_long), ("arrive_cnt", fp), ("arrive_cnt_len", C.c_int)]


class CmmWeights(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("c_img", "cnum", "img_h", "img_w")] + [("en_w", fp * 10), ("en_b", fp * 10)] + [
        (n, fp) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "de6_w", "de6_b")] + [
        ("dea_w", fp * 4), ("dea_b", fp * 4), ("deb_w", fp * 4), ("deb_b", fp * 4), ("de1_w", fp), ("de1_b", fp)]


class PsnSrb(C.Structure):
    _fields_ = [(n, fp) for n in ("c1_w", "c1_b", "c2_w", "c2_b", "g1_w", "g1_b", "g1_whh", "g1_bhh", "g2_w", "g2_b", "g2_whh", "g2_bhh")]


class PsnWeights(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("in_planes", "ch", "hidden", "srb_nums")] + [("srb", PsnSrb * 8)] + [
        (n, fp) for n in ("b7_w", "b7_b", "up_w", "up_b", "last_w", "last_b")]


class TattDecLayer(C.Structure):
    _fields_ = [(n, fp) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "out_w", "out_b", "norm2_w", "norm2_b", "lin1_w", "lin1_b",
                                  "lin2_w", "lin2_b", "norm3_w", "norm3_b")]


class TattInterpWeights(C.Structure):
    _fields_ = [("n_dec", C.c_int), ("nhead", C.c_int), ("fc_in_w", fp), ("fc_in_b", fp), ("fc_in_slope", C.c_float), ("enc", fp * 12),
                ("dec", TattDecLayer * 4), ("dec_norm_w", fp), ("dec_norm_b", fp)]


class TnPending(C.Structure):
    _fields_ = [("part", fp), ("dw", fp), ("db", fp), ("NK", C.c_int), ("N", C.c_int), ("splits", C.c_int)]


class TnItem(C.Structure):
    _fields_ = [("dy", fp), ("x", fp), ("dw", fp), ("db", fp), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ws", fp), ("ws_bytes", C.c_size_t)]


class DistillParams(C.Structure):
    _fields_ = [("conv_cat_w", fp), ("conv_cat_b", fp), ("bn1_w", fp), ("bn1_b", fp), ("bn1_rm", fp), ("bn1_rv", fp), ("bn1_nbt", fp),
                ("conv_feat_w", fp), ("conv_feat_b", fp), ("bn2_w", fp), ("bn2_b", fp), ("bn2_rm", fp), ("bn2_rv", fp), ("bn2_nbt", fp)]


class DistillGrads(C.Structure):
    _fields_ = [("dconv_cat_w", fp), ("dconv_cat_b", fp), ("dbn1_w", fp), ("dbn1_b", fp), ("dconv_feat_w", fp), ("dconv_feat_b", fp),
                ("dbn2_w", fp), ("dbn2_b", fp)]


class CmmScratch(C.Structure):
    _fields_ = [("splitk_ws", fp), ("splitk_ws_bytes", C.c_size_t), ("arrive_cnt", fp), ("arrive_cnt_len", C.c_int)]


_i, _f, _sz, _l, _u64 = C.c_int, C.c_float, C.c_size_t, C.c_long, C.c_ulonglong
_PP = C.POINTER(fp)
_IP = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol include/dpmn_hip.h declares
SIGNATURES = {
    "dpmn_abi_version": (_i, []),
    "dpmn_last_error": (C.c_char_p, []),
    "dpmn_set_compute_dtype": (_i, [_i]),
    "dpmn_get_compute_dtype": (_i, []),
    "dpmn_linear_f32": (_i, [fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _f, fp]),
    "dpmn_add_linear_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_cat2_linear_f32": (_i, [fp, _i, fp, _i, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_ln_linear_f32": (_i, [fp, fp, fp, _f, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_proj_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, fp]),
    "dpmn_sk_select_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_mlp_in_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_bwd_det_bytes": (_sz, [_i, _i, _i]),
    "dpmn_linear_drop_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _f, _u64, _f, _u64, _l, fp]),
    "dpmn_sk_mlp_in_drop_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_sk_mlp_in_supported": (_i, [_i, _i, _i, _i, _i]),
    "dpmn_pointwise_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_conv2d_nhwc_f32": (_i, [C.POINTER(ConvDesc), fp]),
    "dpmn_distill_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpmn_distill_forward_f32": (_i, [C.POINTER(DistillParams), fp, fp, _i, fp, fp, fp, fp, fp, _sz, _i, _i, _i, fp]),
    "dpmn_distill_backward_f32": (_i, [C.POINTER(DistillParams), C.POINTER(DistillGrads), fp, fp, fp, fp, fp, fp, fp, fp, fp, _sz, _i, _i, _i, fp]),
    "dpmn_conv2d_wgrad_unpack_multi_f32": (_i, [fp, fp, _i, _i, fp]),
    "dpmn_tl_interp_f32": (_i, [fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_pointwise_wgrad_det_bytes": (_sz, [_i, _i]),
    "dpmn_reduce_defer_begin": (_i, []),
    "dpmn_reduce_defer_enable": (_i, [_i]),
    "dpmn_reduce_defer_push": (_i, [C.POINTER(TnPending)]),
    "dpmn_reduce_defer_pending": (_i, []),
    "dpmn_reduce_defer_flush": (_i, [_i, fp]),
    "dpmn_xred_fallbacks": (_i, [C.POINTER(C.c_uint), _i]),
    "dpmn_selftest_xshfl": (_i, [C.POINTER(C.c_uint)]),
    "dpmn_selftest_lds_poison": (_i, [C.c_uint, _i, fp]),
    "dpmn_xred_test_force_recompute": (_i, [_i]),
    "dpmn_xred_enable": (_i, [_i]),
    "dpmn_nchw_to_nhwc_f32": (_i, [fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_nhwc_to_nchw_f32": (_i, [fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_se_gate_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_bigru_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, C.c_long, C.c_long, C.c_long, _i, fp]),
    "dpmn_small_linear_f32": (_i, [fp, fp, _i, fp, fp, fp, _i, _i, _i, _i, _f, fp]),
    "dpmn_tatt_encoder_layer_f32": (_i, [fp, fp, _PP, fp, _i, _i, _i, _i, fp]),
    "dpmn_cross_attn_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_add_layernorm64_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, _f, _i, C.c_long, fp]),
    "dpmn_gru_gate_f32": (_i, [fp, fp, fp, fp, C.c_long, _i, _i, fp]),
    "dpmn_to_mask_f32": (_i, [fp, C.c_long, fp, _i, _i, _i, fp]),
    "dpmn_mha32_f32": (_i, [fp, fp, _i, _i, _i, _f, fp]),
    "dpmn_layernorm_std_f32": (_i, [fp, fp, fp, _f, fp, _l, _i, fp]),
    "dpmn_rotate_img_f32": (_i, [fp, fp, fp, _f, fp, _i, _i, _i, _i, fp]),
    "dpmn_blend_f32": (_i, [fp, C.c_long, fp, C.c_long, fp, _f, _i, _i, fp]),
    "dpmn_psnr_ssim_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpmn_psnr_ssim_f32": (_i, [fp, C.c_long, fp, C.c_long, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_gemm_tn_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp, _sz, fp]),
    "dpmn_gemm_tn_partial_bytes": (_sz, [_i, _i, _i]),
    "dpmn_gemm_tn_partial_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp, _sz, C.POINTER(TnPending), fp]),
    "dpmn_tn_reduce_multi_f32": (_i, [C.POINTER(TnPending), _i, fp]),
    "dpmn_gemm_tn_group_f32": (_i, [C.POINTER(TnItem), _i, fp]),
    "dpmn_colsum_f32": (_i, [fp, fp, C.c_long, _i, fp]),
    "dpmn_colsum_det_f32": (_i, [fp, fp, C.c_long, _i, fp, _sz, fp]),
    "dpmn_layernorm_bwd_f32": (_i, [fp, fp, fp, _f, fp, _i, fp, fp, C.c_long, _i, fp]),
    "dpmn_layernorm_bwd_det_f32": (_i, [fp, fp, fp, _f, fp, _i, fp, fp, C.c_long, _i, fp, _sz, fp]),
    "dpmn_layernorm_bwd_det_drop_f32": (_i, [fp, fp, fp, _f, fp, _i, fp, fp, C.c_long, _i, fp, _sz, fp, _f, _u64, _f, _u64, _l, fp]),
    "dpmn_layernorm_f32": (_i, [fp, fp, fp, _f, fp, C.c_long, _i, fp]),
    "dpmn_act_bwd_f32": (_i, [fp, fp, fp, _i, _f, C.c_long, fp]),
    "dpmn_act_fwd_f32": (_i, [fp, fp, _i, _f, C.c_long, fp]),
    "dpmn_axpby_f32": (_i, [fp, fp, fp, _f, _f, _i, C.c_long, fp]),
    "dpmn_rowsum_mod_f32": (_i, [fp, fp, C.c_long, _i, _i, fp]),
    "dpmn_rows_reduce_f32": (_i, [fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_rowsum_mod_det_f32": (_i, [fp, fp, C.c_long, _i, _i, fp, _sz, fp]),
    "dpmn_dwconv3x3_bwd_fused_det_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, _i, _i, _f, _u64, _i, _i, _i, fp, _sz, fp]),
    "dpmn_image_loss_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dpmn_image_loss_fwd_f32": (_i, [fp, C.c_long, fp, C.c_long, _f, _f, _i, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_image_loss_bwd_f32": (_i, [fp, C.c_long, fp, C.c_long, fp, fp, fp, _f, _f, _i, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_window_attn_bwd_f32": (_i, [fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, fp, _PP, _i, _i, _i, _i, fp]),
    "dpmn_sk_select_only_f32": (_i, [fp, fp, fp, C.c_long, _i, _i, _i, fp]),
    "dpmn_sk_select_bwd_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_gate_bwd_f32": (_i, [fp, _i, _i, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_select_bwd_det_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_select_bwd_det_set_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_gate_bwd_det_f32": (_i, [fp, _i, _i, fp, fp, fp, fp, fp, _i, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_sk_feats_grad_f32": (_i, [fp, fp, fp, fp, C.c_long, _i, _i, fp]),
    "dpmn_dwconv3x3_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_gelu_in_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_bwd_f32": (_i, [fp, fp, fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_train_f32": (_i, [fp, fp, fp, fp, fp, _i, _f, _u64, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_bwd_fused_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, _i, _i, _f, _u64, _i, _i, _i, fp]),
    "dpmn_pointwise_wgrad_f32": (_i, [fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_pointwise_wgrad_det_f32": (_i, [fp, fp, fp, _i, _i, _i, fp, _sz, fp]),
    "dpmn_pgrm_tail_elem_f32": (_i, [fp, _PP, _PP, _i, fp, _i, _i, _i, fp]),
    "dpmn_pgrm_tail_elem_bwd_f32": (_i, [fp, fp, _PP, _PP, _PP, _PP, _i, fp, _i, _i, _i, fp]),
    "dpmn_patch_embed_bwd_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_patch_scatter_f32": (_i, [fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_prior_fusion_wgrad_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_patch_embed_bwd_det_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_patch_embed_bwd_det_drop_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_patch_embed_bwd_det_wgrad_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_prior_fusion_wgrad_det_f32": (_i, [fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_conv2d_wgrad_f32": (_i, [C.POINTER(ConvDesc), fp, fp, _i, fp]),
    "dpmn_conv_pack_f32": (_i, [fp, fp, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, fp]),
    "dpmn_conv2d_wgrad_excl_slots": (_i, [C.POINTER(ConvDesc), fp]),
    "dpmn_conv2d_wgrad_excl_f32": (_i, [C.POINTER(ConvDesc), fp, fp, _i, fp]),
    "dpmn_conv2d_wgrad_unpack_f32": (_i, [fp, fp, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, fp]),
    "dpmn_conv2d_wgrad_strided_f32": (_i, [C.POINTER(ConvDesc), fp, fp, _i, _i, _l, _l, _l, _l, _l, fp]),
    "dpmn_bn_finalize_f32": (_i, [fp, fp, fp, _f, _f, _f, fp, fp, fp, fp, fp, fp, _i, fp, _i, fp]),
    "dpmn_affine_act_bwd_f32": (_i, [fp, fp, fp, fp, _i, fp, _i, C.c_long, _i, fp]),
    "dpmn_bn_bwd_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_long, _i, fp]),
    "dpmn_affine_act_bwd_stats_f32": (_i, [fp, fp, fp, fp, _i, fp, _i, C.c_long, _i, fp, fp, fp, fp]),
    "dpmn_bn_bwd_apply_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_long, _i, fp]),
    "dpmn_se_gate_bwd_f32": (_i, [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_affine_act_fwd_f32": (_i, [fp, fp, fp, _i, fp, C.c_long, _i, fp]),
    "dpmn_l1_loss_fwd_f32": (_i, [fp, fp, _f, fp, fp, C.c_long, fp]),
    "dpmn_l1_loss_bwd_f32": (_i, [fp, fp, fp, _f, fp, fp, fp, C.c_long, fp]),
    "dpmn_sumsq_f32": (_i, [fp, fp, fp, C.c_long, fp]),
    "dpmn_adam_clip_f32": (_i, [fp, fp, fp, fp, fp, _f, _f, _f, _f, _f, _i, fp, C.c_long, fp]),
    "dpmn_patch_embed_ln_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_patch_embed_ln_drop_f32": (_i, [fp, _i, fp, fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_window_attn_f32": (_i, [fp, fp, _PP, _IP, _IP, _i, _i, fp, _i, _i, _i, _i, fp]),
    "dpmn_window_attn_drop_f32": (_i, [fp, fp, _PP, _IP, _IP, _i, _i, fp, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_window_attn_drop_bwd_f32": (_i, [fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, fp, _PP, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_window_attn_drop_bwd_det_f32": (_i, [fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, fp, _PP, _IP, _i, _i, _i, _i, _f, _u64, fp]),
    "dpmn_window_attn_bwd_part_rows": (_i, [_i, _i, _i]),
    "dpmn_conv_pack_multi_f32": (_i, [fp, fp, _i, _i, fp]),
    "dpmn_conv_pack_tile_shape": (_i, [_i, _i, _i, _l, _l, fp]),
    "dpmn_mha64_f32": (_i, [fp, fp, _i, _i, _i, _f, fp]),
    "dpmn_vl_resize_f32": (_i, [fp, _l, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_vl_tokens_f32": (_i, [fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_vl_pp_pool_f32": (_i, [fp, _i, fp, fp, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_vl_decode_i32": (_i, [fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_text_prior_compose_f32": (_i, [fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, _i, fp]),
    "dpmn_ln_qkv_window_attn_supported": (_i, [_i, _i, _i, _IP, _i, _i]),
    "dpmn_ln_qkv_window_attn_f32": (_i, [fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_ln_qkv_window_attn_train_f32": (_i, [fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, fp, _f, _u64, fp, _i, _i, _i, _i, fp]),
    "dpmn_ln_qkv_window_attn_workspace_bytes": (_sz, []),
    "dpmn_ln_qkv_window_attn_d32_supported": (_i, [_i, _i, _i, _IP, _i, _i]),
    "dpmn_ln_qkv_window_attn_d32_workspace_bytes": (_sz, []),
    "dpmn_ln_qkv_window_attn_d32_f32": (_i, [fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, _i, _i, _i, _i, _i, fp]),
    "dpmn_ln_qkv_window_attn_bwd_f32": (_i, [fp, fp, fp, fp, fp, fp, _f, fp, fp, fp, fp, _PP, _IP, _IP, _i, _i, fp, fp, fp, _PP, _f, _u64, fp, _i,
                                              _i, _i, _i, _i, fp]),
    "dpmn_ln_qkv_window_attn_bwd_part_rows": (_i, [_i, _i, _i]),
    "dpmn_collate_u8_f32": (_i, [fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_profile_tag_count": (_i, []),
    "dpmn_profile_hint_bytes": (_i, [C.c_double]),
    "dpmn_profile_tag_name": (C.c_char_p, [_i]),
    "dpmn_profile_begin": (_i, [C.c_ulonglong, _i]),
    "dpmn_profile_end": (_i, [C.c_void_p, _i]),
    "dpmn_dropout_f32": (_i, [fp, fp, fp, _l, _l, _f, _u64, _f, _u64, fp]),
    "dpmn_maxpool_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "dpmn_stn_fc_f32": (_i, [fp, fp, fp, _i, fp, fp, fp, fp, fp, fp, _i, _f, _f, fp, fp, fp, fp, _i, _i, fp]),
    "dpmn_tps_sample_f32": (_i, [fp, fp, fp, fp, fp, fp, _i, _i, _i, _i, _i, _i, _i, fp]),
    "dpmn_sk_gate_f32": (_i, [fp, _i, _i, fp, fp, fp, fp, fp, _i, _i, _i, _i, fp]),
    "dpmn_dwconv3x3_gelu_f32": (_i, [fp, fp, fp, fp, _i, _i, _i, fp]),
    "dpmn_pgrm_tail_f32": (_i, [fp, fp, fp, fp, fp, _PP, _PP, _i, fp, fp, _i, _i, _i, _i, _i, _i, fp]),
    "dpmn_pgrm_tail_reuse_f32": (_i, [fp, fp, fp, fp, fp, _PP, _PP, _i, fp, fp, _i, _i, _i, _i, _i, _i, _i, fp]),
    "dpmn_tatt_interpreter_workspace_bytes": (_sz, [_i, _i, _i]),
    "dpmn_tatt_interpreter_f32": (_i, [C.POINTER(TattInterpWeights), fp, _i, fp, fp, fp, fp, fp, fp, _sz, _i, _i, _i, fp]),
    "dpmn_psn_trunk_workspace_bytes": (_sz, [C.POINTER(PsnWeights), _i, _i, _i]),
    "dpmn_psn_trunk_f32": (_i, [C.POINTER(PsnWeights), fp, fp, _i, fp, fp, _sz, C.POINTER(CmmScratch), _i, _i, _i, fp]),
    "dpmn_cmm_workspace_bytes": (_sz, [C.POINTER(CmmWeights), _i]),
    "dpmn_cmm_forward_f32": (_i, [C.POINTER(CmmWeights), fp, fp, fp, fp, _sz, C.POINTER(CmmScratch), _i, fp]),
    "dpmn_pgrm_workspace_bytes": (_sz, [C.POINTER(PgrmWeights), _i]),
    "dpmn_pgrm_forward_f32": (_i, [C.POINTER(PgrmWeights), fp, _i, fp, _PP, _i, fp, fp, _sz, _i, fp]),
    "dpmn_pgrm_forward_train_supported": (_i, [C.POINTER(PgrmWeights), _i]),
    "dpmn_pgrm_blocks_backward_scratch_bytes": (_sz, [C.POINTER(PgrmWeights), _i, C.POINTER(C.c_int)]),
    "dpmn_pgrm_blocks_backward_f32": (_i, [C.POINTER(PgrmWeights), C.POINTER(PgrmBlock), C.POINTER(PgrmBlockT), C.POINTER(PgrmSaved),
                                           C.POINTER(PgrmDrop), C.POINTER(C.c_int), fp, fp, _PP, fp, fp, _sz, fp, _sz, C.POINTER(C.c_size_t),
                                           _i, fp]),
    "dpmn_pgrm_blocks_backward_leaf_f32": (_i, [C.POINTER(PgrmWeights), C.POINTER(PgrmBlock), C.POINTER(PgrmBlockT), C.POINTER(PgrmSaved),
                                                C.POINTER(PgrmDrop), C.POINTER(C.c_int), fp, fp, _PP, fp, fp, _sz, fp, _sz, C.POINTER(C.c_size_t),
                                                _i, fp, fp]),
    "dpmn_pgrm_forward_train_f32": (_i, [C.POINTER(PgrmWeights), fp, _i, fp, _PP, _i, fp, fp, C.POINTER(PgrmDrop), C.POINTER(PgrmSaved),
                                         C.POINTER(CmmScratch), fp, _i, fp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:  # pragma: no cover
        raise RuntimeError("dpmn_amd: %s does not export %s (stale build?)" % (LIB_PATH, _name)) from e
    _fn.restype = _res
    _fn.argtypes = _args


class DpmnError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise DpmnError("libdpmn_hip error %d: %s" % (rc, lib.dpmn_last_error().decode()))


# DPMN_COMPUTE_DTYPE = f32 | bf16 | x3 (or 0 | 1 | 2): the process-wide arithmetic mode of the MFMA kernels that have variants
# (dpmn_set_compute_dtype, include/dpmn_hip.h) for tools and whole-suite test passes; the default is fp32 MFMA
_mode = os.environ.get("DPMN_COMPUTE_DTYPE")
if _mode:
    check(lib.dpmn_set_compute_dtype({"f32": 0, "bf16": 1, "x3": 2}[_mode] if _mode in ("f32", "bf16", "x3") else int(_mode)))


def dptr(t, allow_none=False):
    """Device pointer of a contiguous fp32 CUDA tensor (the only thing the C ABI accepts)."""
    if t is None:
        if allow_none:
            return None
        raise DpmnError("dpmn_amd: required tensor is None")
    if not t.is_cuda:
        raise DpmnError("dpmn_amd: the DPMN hot path runs on the GPU only (got a %s tensor); "
                        "there is no CPU fallback" % t.device)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise DpmnError("dpmn_amd: expected a contiguous float32 tensor, got %s %s" % (t.dtype, tuple(t.stride())))
    return t.data_ptr()


def ptr_array(tensors, n=None):
    n = len(tensors) if n is None else n
    arr = (fp * max(n, 1))()
    for i, t in enumerate(tensors):
        arr[i] = dptr(t)
    return arr


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream():
    """hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object: ~7 us per
    call, times ~1400 launches per training step -- a third of the step's host time; the raw query is ~0.3 us.)"""
    if _raw_stream is not None:
        return _raw_stream(_get_device())       # (torch.cuda.current_device() = _lazy_init() + this call)
    return torch.cuda.current_stream().cuda_stream


def stream_of(device):
    """hipStream_t of torch's current stream on `device`, as an integer (workspace-cache keys); same raw query as stream()."""
    if _raw_stream is not None:
        idx = getattr(device, "index", None)
        return _raw_stream(_get_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


# ------------------------------------------------------------------ in-pipeline kernel profiler (dpmn_profile_*)
class ProfileRow(C.Structure):
    _fields_ = [("tag", C.c_int), ("launches", C.c_int), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


def profile_tags():
    return [lib.dpmn_profile_tag_name(i).decode() for i in range(lib.dpmn_profile_tag_count())]


def profile_begin(tags=None, max_launches=1 << 16):
    """Arm the library's launch timers for the named kernel families (None = all)."""
    names = profile_tags()
    mask = 0
    for i, n in enumerate(names):
        if tags is None or n in tags:
            mask |= 1 << i
    check(lib.dpmn_profile_begin(mask, max_launches))


def profile_end():
    """Disarm; returns [dict(kernel, launches, total_ms, flops, bytes)].  The stream must be synchronised."""
    n_tags = lib.dpmn_profile_tag_count()
    rows = (ProfileRow * n_tags)()
    n = lib.dpmn_profile_end(C.cast(rows, C.c_void_p), n_tags)
    if n < 0:
        raise DpmnError(lib.dpmn_last_error().decode())
    names = profile_tags()
    return [dict(kernel=names[r.tag], launches=r.launches, total_ms=r.total_ms, flops=r.flops, bytes=r.bytes) for r in rows[:n]]
